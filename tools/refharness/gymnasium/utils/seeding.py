import numpy as np


def np_random(seed=None):
    """Mirrors gymnasium's documented construction: Generator(PCG64(SeedSequence(seed)))."""
    seed_seq = np.random.SeedSequence(seed)
    np_seed = seed_seq.entropy
    rng = np.random.Generator(np.random.PCG64(seed_seq))
    return rng, np_seed
