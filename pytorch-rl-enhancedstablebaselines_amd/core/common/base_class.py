"""`BaseAlgorithm`: constructor contract, env wrapping, seeding, lr schedule, learn-time bookkeeping
(reference: core/common/base_class.py:69-889; only what the off-policy CSTR path touches), and save/load in the
reference's zip layout (core/common/save_util.py)."""
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
from core.common.vec_env import CSTRVecEnv, VecEnv, unwrap_vec_normalize


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
            self._vec_normalize_env = unwrap_vec_normalize(env)  # reference base_class.py:195
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
        self._reseed_device_rng(seed)

    def _reseed_device_rng(self, seed: int) -> None:
        """Device-resident counter RNG streams of the fused kernels follow `set_random_seed` like torch's generator does."""
        fa = getattr(self, "_fast_actor", None)
        if fa is not None and hasattr(fa, "seed_rng"):
            fa.seed_rng(seed + 1000003 * self.rank)
        ctl = getattr(self, "_rng_ctl", None)  # target-smoothing noise stream (TD3 / MADDPG)
        if ctl is not None:
            from core.common import hip_ops

            ctl.copy_(hip_ops.new_rng_ctl(seed + 1000003 * self.rank, ctl.device))

    def _device_rng(self):
        """In-kernel Philox stream for this algorithm's own noise kernels, seeded from torch's seed at first use."""
        if getattr(self, "_rng_ctl", None) is None:
            from core.common import hip_ops

            self._rng_ctl = hip_ops.new_rng_ctl(th.initial_seed(), self.device)
        return self._rng_ctl

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
        self._vec_normalize_env = unwrap_vec_normalize(env)  # reference base_class.py:525

    def get_vec_normalize_env(self):
        """reference: base_class.py:491-498"""
        return self._vec_normalize_env

    @property
    def _denv(self) -> Optional[CSTRVecEnv]:
        """the device-resident env underneath `self.env` (itself, or the one a VecNormalize wraps); None otherwise"""
        inner = getattr(self.env, "unwrapped", None)
        return inner if isinstance(inner, CSTRVecEnv) else None

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
            self._last_obs = self.env.reset_device() if self._denv is not None else self.env.reset()
            self._last_episode_starts = np.ones((self.env.num_envs,), dtype=bool)
            # a seeded TwoSeriesCSTREnv.reset re-seeds the GLOBAL numpy stream (twoseriescstr.py:164): the replay
            # sampler's stream ends up seeded with seed + n_envs - 1 (SURVEY a-6)
            reseed = getattr(self._denv or self.env, "numpy_reseed", None)
            if reseed is not None:
                from core.common import legacy_rng

                np.random.seed(reseed)
                legacy_rng.seed(reseed, self.device)
                (self._denv or self.env).numpy_reseed = None
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

    # ---- checkpoints (reference: base_class.py:666-888, save_util.py:294-466) --------------------------------------
    _SAVE_HYPERS = ("learning_rate", "buffer_size", "learning_starts", "batch_size", "tau", "gamma", "gradient_steps", "seed",
                    "policy_kwargs", "num_timesteps", "_n_updates", "_episode_num", "n_envs", "_total_timesteps",
                    "_num_timesteps_at_start", "_current_progress_remaining", "verbose")

    def _get_torch_save_params(self) -> tuple:
        """(state-dict attribute names, plain tensor attribute names) -- overridden per algorithm like the reference."""
        return ["policy"], []

    def _extra_save_data(self) -> dict:
        return {}

    def _recursive_getattr(self, name: str):
        obj = self
        for part in name.split("."):
            obj = getattr(obj, part)
        return obj

    def get_parameters(self) -> dict:
        """reference: base_class.py:827-840 -- {name: state_dict} for the policy and every optimiser"""
        names, _ = self._get_torch_save_params()
        return {name: self._recursive_getattr(name).state_dict() for name in names}

    def set_parameters(self, load_path_or_dict, exact_match: bool = True, device="auto") -> None:
        """reference: base_class.py:597-664. Values are copied INTO the arena views (module identity is preserved)."""
        if isinstance(load_path_or_dict, dict):
            params = load_path_or_dict
        else:
            from core.common.save_util import load_from_zip_file

            _, params, _ = load_from_zip_file(load_path_or_dict, device="cpu")
        names, _ = self._get_torch_save_params()
        updated = set()
        for name in params:
            if name not in names:
                raise ValueError(f"Key {name} is an invalid object name.")
            target = self._recursive_getattr(name)
            if name == "policy" or isinstance(target, th.nn.Module):
                own = target.state_dict()
                if exact_match and set(own) != set(params[name]):
                    raise ValueError(f"Names of parameters do not match agents' parameters: expected {sorted(own)}, got {sorted(params[name])}")
                with th.no_grad():
                    for k, v in params[name].items():
                        if k in own:
                            own[k].copy_(th.as_tensor(v).to(own[k].device))
            else:
                target.load_state_dict(params[name])
            updated.add(name)
        if exact_match and updated != set(names):
            raise ValueError(f"Names of parameters do not match agents' parameters: expected {names}, got {sorted(updated)}")

    def save(self, path, exclude=None, include=None) -> None:
        """reference: base_class.py:842-888. Writes the same archive members; `data` holds the JSON-able constructor
        arguments and counters (nothing is pickled)."""
        from core import __version__
        from core.common.save_util import save_to_zip_file

        data = {k: getattr(self, k) for k in self._SAVE_HYPERS if hasattr(self, k)}
        tf = getattr(self, "train_freq", None)
        if tf is not None and hasattr(tf, "unit"):
            data["train_freq"] = [tf.frequency, tf.unit.value]
        data["algo"] = type(self).__name__
        for nm, sp in (("observation_space", self.observation_space), ("action_space", self.action_space)):
            data[nm] = {"low": np.asarray(sp.low).tolist(), "high": np.asarray(sp.high).tolist(), "shape": list(sp.shape), "dtype": str(sp.dtype)}
        data.update(self._extra_save_data())
        for k in (exclude or []):
            data.pop(k, None)
        _, var_names = self._get_torch_save_params()
        variables = {n: self._recursive_getattr(n).detach().clone() for n in var_names} if var_names else None
        save_to_zip_file(path, data=data, params=self.get_parameters(), pytorch_variables=variables, version=__version__)

    @classmethod
    def load(cls, path, env=None, device="auto", custom_objects=None, print_system_info: bool = False, force_reset: bool = True, **kwargs):
        """reference: base_class.py:666-825. Works for archives written by this stack and for archives written by the
        reference (only `weights_only` tensors and plain JSON are read; `env` must then be passed)."""
        from core.common.save_util import load_from_zip_file

        data, params, variables = load_from_zip_file(path, device="cpu")
        if env is None:
            raise ValueError("load(): pass `env` (environments are not stored in the archive)")
        ctor = {k: data[k] for k in cls._ctor_keys() if k in data}
        if "train_freq" in data and isinstance(data["train_freq"], list):
            ctor["train_freq"] = (int(data["train_freq"][0]), str(data["train_freq"][1]))
        ctor.update(kwargs)
        model = cls._construct_for_load(env, device, ctor)
        model.set_parameters(params, exact_match=True)
        for name, value in (variables or {}).items():
            tgt = model._recursive_getattr(name)
            with th.no_grad():
                tgt.copy_(th.as_tensor(value).to(tgt.device).reshape(tgt.shape))
        for k in ("num_timesteps", "_n_updates", "_episode_num", "_total_timesteps", "_num_timesteps_at_start"):
            if k in data:
                setattr(model, k, data[k])
        return model

    @classmethod
    def _ctor_keys(cls) -> tuple:
        return ("learning_rate", "buffer_size", "learning_starts", "batch_size", "tau", "gamma", "gradient_steps", "seed", "policy_kwargs")

    @classmethod
    def _construct_for_load(cls, env, device, ctor: dict):
        return cls("MlpPolicy", env, device=device, **ctor)
