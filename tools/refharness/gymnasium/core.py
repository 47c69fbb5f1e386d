from typing import Any, Optional, TypeVar

import numpy as np

from .utils import seeding

ObsType = TypeVar("ObsType")
ActType = TypeVar("ActType")


class Env:
    def __class_getitem__(cls, item):
        return cls

    metadata: dict = {"render_modes": []}
    render_mode = None
    spec = None
    observation_space = None
    action_space = None
    _np_random: Optional[np.random.Generator] = None
    _np_random_seed: Optional[int] = None

    def step(self, action):
        raise NotImplementedError

    def reset(self, *, seed: Optional[int] = None, options: Optional[dict] = None):
        if seed is not None:
            self._np_random, self._np_random_seed = seeding.np_random(seed)

    def render(self):
        raise NotImplementedError

    def close(self):
        pass

    @property
    def unwrapped(self):
        return self

    @property
    def np_random(self) -> np.random.Generator:
        if self._np_random is None:
            self._np_random, self._np_random_seed = seeding.np_random()
        return self._np_random

    @np_random.setter
    def np_random(self, value):
        self._np_random = value

    def has_wrapper_attr(self, name: str) -> bool:
        return hasattr(self, name)

    def get_wrapper_attr(self, name: str) -> Any:
        return getattr(self, name)

    def set_wrapper_attr(self, name: str, value: Any):
        setattr(self, name, value)


class Wrapper(Env):
    def __init__(self, env):
        self.env = env

    def __getattr__(self, name):
        if name.startswith("_"):
            raise AttributeError(name)
        return getattr(self.env, name)

    @property
    def observation_space(self):
        return self.env.observation_space

    @property
    def action_space(self):
        return self.env.action_space

    @property
    def unwrapped(self):
        return self.env.unwrapped

    def step(self, action):
        return self.env.step(action)

    def reset(self, *, seed=None, options=None):
        return self.env.reset(seed=seed, options=options)

    def close(self):
        return self.env.close()

    def get_wrapper_attr(self, name):
        if name in self.__dict__ or hasattr(type(self), name):
            return getattr(self, name)
        return self.env.get_wrapper_attr(name)


class ObservationWrapper(Wrapper):
    pass


class RewardWrapper(Wrapper):
    pass


class ActionWrapper(Wrapper):
    pass
