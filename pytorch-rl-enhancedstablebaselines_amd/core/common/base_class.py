"""`BaseAlgorithm`: constructor contract, env wrapping, seeding, lr schedule, learn-time bookkeeping
(reference: core/common/base_class.py:69-889; only what the off-policy CSTR path touches).
save/load (zip checkpoints) are a "next" row and raise NotImplementedError for now."""
import time
from collections import deque
from typing import Any, Optional, Union

import numpy as np
import torch as th

from core.common import distributed as dist_util
from core.common.callbacks import BaseCallback, MaybeCallback, to_callback
from core.common.logger import Logger, configure_logger
from core.common.spaces import Box, as_box
from core.common.utils import get_device, get_schedule_fn, set_random_seed, update_learning_rate
from core.common.vec_env import CSTRVecEnv, VecEnv


class BaseAlgorithm:
    policy_aliases: dict = {}

    def __init__(self, policy, env, learning_rate, policy_kwargs: Optional[dict] = None, stats_window_size: int = 100,
                 tensorboard_log: Optional[str] = None, verbose: int = 0, device: Union[th.device, str] = "auto",
                 support_multi_env: bool = False, monitor_wrapper: bool = True, seed: Optional[int] = None,
                 use_sde: bool = False, sde_sample_freq: int = -1, supported_action_spaces: Optional[tuple] = None):
        if isinstance(policy, str):
            self.policy_class = self._get_policy_from_name(policy)
        else:
            self.policy_class = policy
        self.device = get_device(device)
        self.verbose = verbose
        self.policy_kwargs = {} if policy_kwargs is None else policy_kwargs
        self.num_timesteps = 0
        self._total_timesteps = 0
        self._num_timesteps_at_start = 0
        self.seed = seed
        self.action_noise = None
        self.start_time = 0
        self.learning_rate = learning_rate
        self.tensorboard_log = tensorboard_log
        self._last_obs = None
        self._last_episode_starts = None
        self._last_original_obs = None
        self._episode_num = 0
        if use_sde:
            raise NotImplementedError("gSDE is out of scope for the CSTR path (SURVEY 2)")
        self.use_sde, self.sde_sample_freq = use_sde, sde_sample_freq
        self._current_progress_remaining = 1.0
        self._stats_window_size = stats_window_size
        self.ep_info_buffer: Optional[deque] = None
        self.ep_success_buffer: Optional[deque] = None
        self._n_updates = 0
        self._custom_logger = False
        self._logger: Optional[Logger] = None
        self._vec_normalize_env = None
        self.rank, self.world_size = dist_util.rank_world()
        self.env: Optional[VecEnv] = None
        self.policy = None
        if env is not None:
            env = self._wrap_env(env, self.verbose)
            self.observation_space = as_box(env.observation_space)
            self.action_space = as_box(env.action_space)
            self.n_envs = env.num_envs
            self.env = env
            if supported_action_spaces is not None and not isinstance(self.action_space, Box):
                raise AssertionError(f"The algorithm only supports {supported_action_spaces} as action spaces")
            if not support_multi_env and self.n_envs > 1:
                raise ValueError("Error: the model does not support multiple envs; it requires a single vectorized environment.")
            # reference base_class.py:215-218
            assert np.all(np.isfinite(np.array([self.action_space.low, self.action_space.high]))), \
                "Continuous action space must have a finite lower and upper bound"

    # ---- env / policy plumbing --------------------------------------------------------------------------------
    @staticmethod
    def _wrap_env(env, verbose: int = 0) -> VecEnv:
        """reference: base_class.py:220-253. A VecEnv passes through untouched; a bare TwoSeriesCSTREnv becomes a
        1-env device CSTRVecEnv (the reference would wrap it in Monitor + DummyVecEnv)."""
        if isinstance(env, VecEnv):
            return env
        from twoseriescstr import TwoSeriesCSTREnv

        if isinstance(env, TwoSeriesCSTREnv):
            return CSTRVecEnv(1, **env.vec_kwargs())
        if all(hasattr(env, a) for a in ("num_envs", "observation_space", "action_space", "reset", "step")):
            return env  # duck-typed VecEnv (compatibility path)
        raise ValueError(f"Unsupported environment {type(env).__name__}: pass a CSTRVecEnv, a DummyVecEnv of "
                         "TwoSeriesCSTREnv or a TwoSeriesCSTREnv")

    def _get_policy_from_name(self, policy_name: str):
        """reference: base_class.py:345-360"""
        if policy_name in self.policy_aliases:
            return self.policy_aliases[policy_name]
        raise ValueError(f"Policy {policy_name} unknown")

    def _setup_lr_schedule(self) -> None:
        self.lr_schedule = get_schedule_fn(self.learning_rate)

    def _update_current_progress_remaining(self, num_timesteps: int, total_timesteps: int) -> None:
        self._current_progress_remaining = 1.0 - float(num_timesteps) / float(total_timesteps)

    def _update_learning_rate(self, optimizers) -> None:
        """reference: base_class.py:303-317"""
        self.logger.record("train/learning_rate", self.lr_schedule(self._current_progress_remaining))
        if not isinstance(optimizers, list):
            optimizers = [optimizers]
        for optimizer in optimizers:
            update_learning_rate(optimizer, self.lr_schedule(self._current_progress_remaining))
            if hasattr(optimizer, "sync_lr"):
                optimizer.sync_lr()

    def set_random_seed(self, seed: Optional[int] = None) -> None:
        """reference: base_class.py:582-595"""
        if seed is None:
            return
        set_random_seed(seed, using_cuda=True, device=self.device)
        self.action_space.seed(seed)
        if self.env is not None:
            self.env.seed(seed)

    def get_env(self):
        return self.env

    def set_env(self, env, force_reset: bool = True) -> None:
        env = self._wrap_env(env, self.verbose)
        if env.num_envs != self.n_envs:
            raise AssertionError("The number of environments to be set is different from the number of environments in the model")
        if as_box(env.observation_space) != self.observation_space or as_box(env.action_space) != self.action_space:
            raise ValueError("Observation/action spaces do not match")
        if force_reset:
            self._last_obs = None
        self.env = env

    @property
    def logger(self) -> Logger:
        if self._logger is None:
            self._logger = configure_logger(self.verbose)
        return self._logger

    def set_logger(self, logger: Logger) -> None:
        self._logger = logger
        self._custom_logger = True

    def _init_callback(self, callback: MaybeCallback) -> BaseCallback:
        callback = to_callback(callback)
        callback.init_callback(self)
        return callback

    # ---- learn-time bookkeeping ---------------------------------------------------------------------------------
    def _setup_learn(self, total_timesteps: int, callback: MaybeCallback = None, reset_num_timesteps: bool = True,
                     tb_log_name: str = "run", progress_bar: bool = False):
        """reference: base_class.py:406-460"""
        self.start_time = time.time_ns()
        if self.ep_info_buffer is None or reset_num_timesteps:
            self.ep_info_buffer = deque(maxlen=self._stats_window_size)
            self.ep_success_buffer = deque(maxlen=self._stats_window_size)
        if self.action_noise is not None:
            self.action_noise.reset()
        if reset_num_timesteps:
            self.num_timesteps = 0
            self._episode_num = 0
        else:
            total_timesteps += self.num_timesteps
        self._total_timesteps = total_timesteps
        self._num_timesteps_at_start = self.num_timesteps
        if reset_num_timesteps or self._last_obs is None:
            assert self.env is not None
            self._last_obs = self.env.reset_device() if isinstance(self.env, CSTRVecEnv) else self.env.reset()
            self._last_episode_starts = np.ones((self.env.num_envs,), dtype=bool)
            # a seeded TwoSeriesCSTREnv.reset re-seeds the GLOBAL numpy stream (twoseriescstr.py:164): the replay
            # sampler's stream ends up seeded with seed + n_envs - 1 (SURVEY a-6)
            reseed = getattr(self.env, "numpy_reseed", None)
            if reseed is not None:
                from core.common import legacy_rng

                np.random.seed(reseed)
                legacy_rng.seed(reseed, self.device)
                self.env.numpy_reseed = None
        if not self._custom_logger:
            self._logger = configure_logger(self.verbose, self.tensorboard_log, tb_log_name, reset_num_timesteps)
        callback = self._init_callback(callback)
        return total_timesteps, callback

    def predict(self, observation, state=None, episode_start=None, deterministic: bool = False):
        """reference: base_class.py:560-580"""
        return self.policy.predict(observation, state, episode_start, deterministic)

    def learn(self, total_timesteps: int, callback: MaybeCallback = None, log_interval: int = 100, tb_log_name: str = "run",
              reset_num_timesteps: bool = True, progress_bar: bool = False):
        raise NotImplementedError

    def train(self, *args: Any, **kwargs: Any) -> None:
        raise NotImplementedError

    def get_parameters(self) -> dict:
        """reference: base_class.py:827-840 (state dicts of the policy and optimisers)"""
        return {"policy": {k: v.detach().clone() for k, v in self.policy.state_dict().items()}}

    def set_parameters(self, load_path_or_dict: dict, exact_match: bool = True, device="auto") -> None:
        if not isinstance(load_path_or_dict, dict):
            raise NotImplementedError("zip checkpoints are a 'next' row (SURVEY 8f-2); pass a dict of state dicts")
        sd = load_path_or_dict["policy"]
        with th.no_grad():  # copy INTO the arena views (load_state_dict would keep them too, but be explicit)
            own = self.policy.state_dict()
            if exact_match and set(own) != set(sd):
                raise ValueError(f"Names of parameters do not match agents' parameters: expected {sorted(own)}, got {sorted(sd)}")
            for k, v in sd.items():
                own[k].copy_(th.as_tensor(v).to(own[k].device))

    def save(self, path, exclude=None, include=None) -> None:
        raise NotImplementedError("SB3 zip checkpoints are a 'next' row (SURVEY 8f-2)")

    @classmethod
    def load(cls, path, env=None, device="auto", **kwargs):
        raise NotImplementedError("SB3 zip checkpoints are a 'next' row (SURVEY 8f-2)")
