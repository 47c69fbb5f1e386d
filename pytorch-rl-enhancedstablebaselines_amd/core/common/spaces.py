"""Minimal `Box` space (the only space on the CSTR path). gymnasium is not a dependency of this stack:
any object with `low`, `high`, `shape`, `dtype` (e.g. a real gymnasium.spaces.Box) is accepted wherever a
Box is expected; `as_box` normalises it. Mirrors the attributes the reference reads
(core/common/preprocessing.py:152-197, core/common/base_class.py:215-218)."""
from typing import Optional, Sequence

import numpy as np


class Space:
    def __init__(self, shape, dtype):
        self._shape = tuple(int(s) for s in shape)
        self.dtype = np.dtype(dtype)
        self._np_random: Optional[np.random.Generator] = None

    @property
    def shape(self):
        return self._shape

    def seed(self, seed: Optional[int] = None):
        ss = np.random.SeedSequence(seed)
        self._np_random = np.random.Generator(np.random.PCG64(ss))
        return [ss.entropy]

    @property
    def np_random(self) -> np.random.Generator:
        if self._np_random is None:
            self.seed()
        return self._np_random


class Box(Space):
    def __init__(self, low, high, shape: Optional[Sequence[int]] = None, dtype=np.float32, seed=None):
        dtype = np.dtype(dtype)
        if shape is None:
            shape = np.shape(low) if not np.isscalar(low) else np.shape(high)
        shape = tuple(int(s) for s in shape)
        self.low = np.full(shape, low, dtype=dtype) if np.isscalar(low) else np.array(low, dtype=dtype)
        self.high = np.full(shape, high, dtype=dtype) if np.isscalar(high) else np.array(high, dtype=dtype)
        if self.low.shape != shape or self.high.shape != shape:
            raise ValueError(f"low/high shape mismatch: {self.low.shape} {self.high.shape} vs {shape}")
        super().__init__(shape, dtype)
        if seed is not None:
            self.seed(seed)

    def sample_batch(self, n: int) -> np.ndarray:
        """`np.array([space.sample() for _ in range(n)])` (off_policy_algorithm.py:388) in one call: n sequential
        bounded-Box draws from one Generator fill row-major, i.e. one uniform(size=(n, *shape)) draw.
        (gymnasium's own Box.sample is not available here: UNPINNED, see DESIGN.md.)"""
        u = self.np_random.uniform(low=self.low, high=self.high, size=(n, *self.shape))
        return u.astype(self.dtype)

    def sample(self) -> np.ndarray:
        return self.sample_batch(1)[0]

    def contains(self, x) -> bool:
        x = np.asarray(x)
        return bool(x.shape == self.shape and np.all(x >= self.low) and np.all(x <= self.high))

    def is_bounded(self) -> bool:
        return bool(np.all(np.isfinite(self.low)) and np.all(np.isfinite(self.high)))

    def __eq__(self, other):
        return (hasattr(other, "low") and hasattr(other, "high") and tuple(other.shape) == self.shape
                and np.array_equal(np.asarray(other.low), self.low) and np.array_equal(np.asarray(other.high), self.high))

    def __repr__(self):
        return f"Box({self.low.min()}, {self.high.max()}, {self.shape}, {self.dtype})"


class IndexedBox(Box):
    """Box that remembers which indices of the global vector it covers
    (reference: core/common/envs/multi_agent_envs.py:7-29)."""

    def __init__(self, low, high, indices, dtype=np.float32):
        super().__init__(low=low, high=high, dtype=dtype)
        self.indices = np.array(indices)

    def map_to_original(self, values):
        values = np.array(values) if isinstance(values, list) else values
        assert values.shape == self.shape, f"value shape {values.shape} does not match space shape {self.shape}"
        return self.indices, values


def split_spaces(observation_space: Box, action_space: Box, observation_splits, action_splits) -> tuple:
    """reference: core/common/envs/multi_agent_envs.py:32-61 -> (obs_subspaces, action_subspaces)"""
    def cut(space, splits):
        out = []
        for idx in splits:
            idx = np.array(idx)
            out.append(IndexedBox(np.asarray(space.low)[idx], np.asarray(space.high)[idx], idx, dtype=space.dtype))
        return out

    return cut(observation_space, observation_splits), cut(action_space, action_splits)


def as_box(space) -> Box:
    if isinstance(space, Box):
        return space
    if all(hasattr(space, a) for a in ("low", "high", "shape", "dtype")):
        return Box(np.asarray(space.low), np.asarray(space.high), tuple(space.shape), space.dtype)
    raise ValueError(f"Unsupported space {space!r}: this stack supports Box observation/action spaces only")


def get_obs_shape(space) -> tuple:
    return tuple(as_box(space).shape)


def get_action_dim(space) -> int:
    return int(np.prod(as_box(space).shape))
