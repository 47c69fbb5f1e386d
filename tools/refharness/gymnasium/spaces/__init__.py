from typing import Optional

import numpy as np

from . import utils  # noqa: F401  (spaces.utils.flatdim)


class Space:
    def __init__(self, shape=None, dtype=None, seed=None):
        self._shape = None if shape is None else tuple(shape)
        self.dtype = None if dtype is None else np.dtype(dtype)
        self._np_random = None
        if seed is not None:
            self.seed(seed)

    def __class_getitem__(cls, item):
        return cls

    @property
    def shape(self):
        return self._shape

    @property
    def np_random(self):
        if self._np_random is None:
            self.seed()
        return self._np_random

    def seed(self, seed: Optional[int] = None):
        ss = np.random.SeedSequence(seed)
        self._np_random = np.random.Generator(np.random.PCG64(ss))
        return [ss.entropy]


class Box(Space):
    def __init__(self, low, high, shape=None, dtype=np.float32, seed=None):
        dtype = np.dtype(dtype)
        if shape is None:
            shape = np.shape(low) if not np.isscalar(low) else np.shape(high)
        shape = tuple(int(s) for s in shape)
        self.low = np.full(shape, low, dtype=dtype) if np.isscalar(low) else np.asarray(low, dtype=dtype)
        self.high = np.full(shape, high, dtype=dtype) if np.isscalar(high) else np.asarray(high, dtype=dtype)
        self.bounded_below = -np.inf < self.low
        self.bounded_above = np.inf > self.high
        super().__init__(shape, dtype, seed)

    def is_bounded(self, manner="both"):
        below, above = bool(np.all(self.bounded_below)), bool(np.all(self.bounded_above))
        return {"both": below and above, "below": below, "above": above}[manner]

    def sample(self):
        # stand-in: bounded case only (uniform); real gymnasium's exact draw order is NOT pinned
        high = self.high if self.dtype.kind == "f" else self.high.astype("int64") + 1
        s = self.np_random.uniform(low=self.low, high=high, size=self.shape)
        return s.astype(self.dtype)

    def contains(self, x):
        x = np.asarray(x)
        return bool(x.shape == self.shape and np.all(x >= self.low) and np.all(x <= self.high))

    def __eq__(self, other):
        return (
            isinstance(other, Box)
            and self.shape == other.shape
            and self.dtype == other.dtype
            and np.allclose(self.low, other.low)
            and np.allclose(self.high, other.high)
        )

    def __repr__(self):
        return f"Box({self.low}, {self.high}, {self.shape}, {self.dtype})"


class Discrete(Space):
    def __init__(self, n, seed=None, start=0):
        self.n, self.start = int(n), int(start)
        super().__init__((), np.int64, seed)


class MultiDiscrete(Space):
    def __init__(self, nvec, dtype=np.int64, seed=None):
        self.nvec = np.asarray(nvec, dtype=dtype)
        super().__init__(self.nvec.shape, dtype, seed)


class MultiBinary(Space):
    def __init__(self, n, seed=None):
        self.n = n
        super().__init__((n,) if np.isscalar(n) else tuple(n), np.int8, seed)


class Dict(Space):
    def __init__(self, spaces=None, seed=None):
        self.spaces = dict(spaces or {})
        super().__init__(None, None, seed)

    def items(self):
        return self.spaces.items()

    def keys(self):
        return self.spaces.keys()

    def values(self):
        return self.spaces.values()


class Tuple(Space):
    def __init__(self, spaces=(), seed=None):
        self.spaces = tuple(spaces)
        super().__init__(None, None, seed)
