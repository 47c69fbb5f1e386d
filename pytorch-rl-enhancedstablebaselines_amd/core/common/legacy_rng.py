"""Device-resident image of NumPy's GLOBAL legacy RandomState (MT19937).

The reference draws replay indices from the process-global stream (`np.random.randint`,
core/common/buffers.py:113,309), which `set_random_seed` seeds (core/common/utils.py:46) and every
seeded `TwoSeriesCSTREnv.reset` re-seeds (twoseriescstr.py:164). This module keeps ONE such stream per
device in HBM (uint32[624] key + pos, stored as int32 bit patterns) so that the sampler kernel consumes
it exactly like NumPy would; `seed()` is `np.random.seed`.
"""
import os
from typing import Dict

import torch as th

from core import _native as nv
from core.common import hip_ops

_streams: Dict[str, th.Tensor] = {}
_last_seed: Dict[str, int] = {}


def global_stream(device) -> th.Tensor:
    device = th.device(device)
    key = str(device)
    if key not in _streams:
        _streams[key] = th.zeros(nv.MT_STATE_WORDS, dtype=th.int32, device=device)
        # an unseeded np.random starts from OS entropy
        seed(int.from_bytes(os.urandom(4), "little"), device)
    return _streams[key]


def seed(value: int, device) -> None:
    """np.random.seed(value) for the device stream (32-bit seeds, like init_genrand)."""
    device = th.device(device)
    if not 0 <= int(value) < 2**32:
        raise ValueError("Seed must be between 0 and 2**32 - 1")  # numpy's message
    key = str(device)
    if key not in _streams:
        _streams[key] = th.zeros(nv.MT_STATE_WORDS, dtype=th.int32, device=device)
    with th.cuda.device(device):
        hip_ops.mt19937_seed(_streams[key], int(value))
    _last_seed[key] = int(value)


def last_seed(device):
    return _last_seed.get(str(th.device(device)))


def new_stream(value: int, device) -> th.Tensor:
    """A private stream (np.random.RandomState(value)), e.g. one per data-parallel shard."""
    st = th.zeros(nv.MT_STATE_WORDS, dtype=th.int32, device=device)
    with th.cuda.device(th.device(device)):
        hip_ops.mt19937_seed(st, int(value))
    return st
