"""VecEnv protocol (reference: core/common/vec_env/base_vec_env.py:50-335); attribute and method names follow it."""
from typing import Any, Optional, Sequence

import numpy as np


class VecEnv:
    """Abstract protocol; attribute and method names follow the reference."""

    def __init__(self, num_envs: int, observation_space, action_space):
        self.num_envs = num_envs
        self.observation_space = observation_space
        self.action_space = action_space
        self.reset_infos: list = [{} for _ in range(num_envs)]
        self._seeds: list = [None for _ in range(num_envs)]
        self._options: list = [{} for _ in range(num_envs)]
        self.render_mode = None

    def reset(self):
        raise NotImplementedError

    def step_async(self, actions) -> None:
        raise NotImplementedError

    def step_wait(self):
        raise NotImplementedError

    def step(self, actions):
        """reference: base_vec_env.py:214-222"""
        self.step_async(actions)
        return self.step_wait()

    def close(self) -> None:
        pass

    def seed(self, seed: Optional[int] = None) -> Sequence[Optional[int]]:
        """reference: base_vec_env.py:292-309 -- env i is seeded with seed + i at its next reset()."""
        if seed is None:
            seed = int(np.random.randint(0, np.iinfo(np.uint32).max, dtype=np.uint32))
        self._seeds = [seed + idx for idx in range(self.num_envs)]
        return self._seeds

    def _reset_seeds(self) -> None:
        self._seeds = [None for _ in range(self.num_envs)]

    def _reset_options(self) -> None:
        self._options = [{} for _ in range(self.num_envs)]

    def get_attr(self, attr_name: str, indices=None) -> list:
        return [getattr(self, attr_name) for _ in self._indices(indices)]

    def set_attr(self, attr_name: str, value: Any, indices=None) -> None:
        setattr(self, attr_name, value)

    def env_method(self, method_name: str, *args, indices=None, **kwargs) -> list:
        return [getattr(self, method_name)(*args, **kwargs) for _ in self._indices(indices)]

    def env_is_wrapped(self, wrapper_class, indices=None) -> list:
        return [False for _ in self._indices(indices)]

    def _indices(self, indices):
        if indices is None:
            return range(self.num_envs)
        if isinstance(indices, int):
            return [indices]
        return indices

    @property
    def unwrapped(self):
        return self
