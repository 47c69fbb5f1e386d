import numpy as np


def flatdim(space):
    return int(np.prod(space.shape))
