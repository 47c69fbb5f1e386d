"""`VecNormalize` for the device-resident CSTR env (reference: core/common/vec_env/vec_normalize.py:15-345,
core/common/running_mean_std.py).

Same constructor, attributes (`obs_rms`, `ret_rms`, `returns`, `training`, `norm_obs`, `norm_reward`, `clip_obs`,
`clip_reward`, `gamma`, `epsilon`) and methods as the reference wrapper, but the running moments live in HBM (f64, like the
reference's) and one HIP launch (`cstr_vecnorm_step_f64`) performs a whole `step_wait`: statistics update, observation and
reward normalisation, discounted-return bookkeeping. The off-policy loop keeps storing ORIGINAL observations / rewards in
the replay ring and normalises sampled batches with the CURRENT statistics (`cstr_vecnorm_apply_f32`), exactly as
`ReplayBuffer._get_samples(env=...)` does in the reference (core/common/buffers.py:143-155, :312-323).
"""
import pickle
from typing import Any, Optional

import numpy as np
import torch as th

from core import _native as nv
from core.common import hip_ops
from core.common.vec_env.base_vec_env import VecEnv
from core.common.vec_env.cstr_vec_env import CSTRVecEnv


class _DeviceMoments:
    """View of one RunningMeanStd (running_mean_std.py:4-55) inside the HBM statistics block: `.mean / .var / .count` read
    (and write) the device words; values are f64 NumPy like the reference's attributes."""

    def __init__(self, state: th.Tensor, mean_at: int, var_at: int, count_at: int, width: Optional[int]):
        self._s, self._m, self._v, self._c, self._w = state, mean_at, var_at, count_at, width

    def _get(self, at):
        x = self._s[at:at + (self._w or 1)].cpu().numpy()
        return x.copy() if self._w else np.float64(x[0])

    def _set(self, at, value):
        self._s[at:at + (self._w or 1)] = th.as_tensor(np.asarray(value, np.float64).reshape(-1), device=self._s.device)

    mean = property(lambda self: self._get(self._m), lambda self, v: self._set(self._m, v))
    var = property(lambda self: self._get(self._v), lambda self, v: self._set(self._v, v))

    @property
    def count(self) -> float:
        return float(self._s[self._c])

    @count.setter
    def count(self, v: float) -> None:
        self._s[self._c] = float(v)


class VecNormalize(VecEnv):
    """reference: vec_normalize.py:15-127 (constructor), :174-246 (step / normalise), :291-345 (reset / save / load)

    :param venv: the `CSTRVecEnv` to wrap (other VecEnvs are outside this stack's scope)
    """

    def __init__(self, venv: CSTRVecEnv, training: bool = True, norm_obs: bool = True, norm_reward: bool = True,
                 clip_obs: float = 10.0, clip_reward: float = 10.0, gamma: float = 0.99, epsilon: float = 1e-8,
                 norm_obs_keys=None):
        if not isinstance(venv, CSTRVecEnv):
            raise ValueError("VecNormalize wraps the device-resident CSTRVecEnv (or DummyVecEnv of TwoSeriesCSTREnv) in this stack")
        if norm_obs_keys is not None:
            raise ValueError("`norm_obs_keys` param is applicable only with `gym.spaces.Dict` observation spaces")  # :118-120
        super().__init__(venv.num_envs, venv.observation_space, venv.action_space)
        self.venv, self.device = venv, venv.device
        self.training, self.norm_obs, self.norm_reward = training, norm_obs, norm_reward
        self.clip_obs, self.clip_reward, self.gamma, self.epsilon = clip_obs, clip_reward, gamma, epsilon
        self.norm_obs_keys = None
        n, d = venv.num_envs, venv.obs_dim
        with th.cuda.device(self.device):
            self._state = th.zeros(nv.VECNORM_STATE_WORDS, dtype=th.float64, device=self.device)
            hip_ops.vecnorm_init(self._state)
            self._returns = th.zeros(n, dtype=th.float64, device=self.device)
            self.norm_obs_dev = th.zeros(n, d, dtype=th.float32, device=self.device)  # what reset()/step() return
            self.norm_rew_dev = th.zeros(n, dtype=th.float32, device=self.device)
        self.obs_rms = _DeviceMoments(self._state, 0, 8, 16, d)
        self.ret_rms = _DeviceMoments(self._state, 17, 18, 19, None)
        self._cfg_key, self._cfg = None, None

    # ---- attribute plumbing ------------------------------------------------------------------------------------
    def __getattr__(self, name: str) -> Any:
        # VecEnvWrapper.__getattr__ (base_vec_env.py:428-450): unknown attributes resolve on the wrapped env
        if name.startswith("__") or name in ("venv",):
            raise AttributeError(name)
        return getattr(self.venv, name)

    @property
    def unwrapped(self) -> CSTRVecEnv:
        return self.venv

    @property
    def returns(self) -> np.ndarray:
        return self._returns.cpu().numpy()

    @property
    def cfg(self) -> nv.VecNormCfg:
        key = (bool(self.training), bool(self.norm_obs), bool(self.norm_reward), float(self.clip_obs), float(self.clip_reward),
               float(self.gamma), float(self.epsilon))
        if key != self._cfg_key:
            self._cfg_key = key
            self._cfg = nv.VecNormCfg(int(key[0]), int(key[1]), int(key[2]), self.venv.obs_dim, *key[3:])
        return self._cfg

    @property
    def cfg_key(self) -> tuple:
        """changes whenever a setting that is baked into captured launches changes (training flag, clips, ...)"""
        self.cfg
        return self._cfg_key

    def env_is_wrapped(self, wrapper_class, indices=None) -> list:
        return [issubclass(VecNormalize, wrapper_class) for _ in self._indices(indices)]

    def seed(self, seed: Optional[int] = None):
        return self.venv.seed(seed)

    def close(self) -> None:
        self.venv.close()

    # ---- device path ---------------------------------------------------------------------------------------------
    def after_device_step(self) -> th.Tensor:
        """The inner env has just stepped (vec step or fused collect): fold its raw obs / reward / done into the statistics
        and refresh the normalised views (vec_normalize.py:174-204)."""
        v = self.venv
        with th.cuda.device(self.device):
            hip_ops.vecnorm_step(self.cfg, self._state, self._returns, v.obs, v._rew, v._done, self.norm_obs_dev, self.norm_rew_dev)
        return self.norm_obs_dev

    def reset_device(self) -> th.Tensor:
        """reference: vec_normalize.py:291-307"""
        obs = self.venv.reset_device()
        with th.cuda.device(self.device):
            hip_ops.vecnorm_step(self.cfg, self._state, self._returns, obs, None, None, self.norm_obs_dev, None)
        return self.norm_obs_dev

    def step_device(self, actions: th.Tensor):
        """(normalised obs, ORIGINAL reward, done, timeout, normalised terminal obs); normalised rewards: `norm_rew_dev`."""
        obs, rew, done, timeout, next_obs = self.venv.step_device(actions)
        self.after_device_step()
        term = next_obs.clone()
        with th.cuda.device(self.device):
            hip_ops.vecnorm_apply(self.cfg, self._state, term, None, None)  # terminal observations (:198-202)
        return self.norm_obs_dev, rew, done, timeout, term

    def normalize_batch_(self, obs: th.Tensor, next_obs: th.Tensor, rewards: th.Tensor) -> None:
        """ReplayBuffer._get_samples(env=self) on device tensors, in place (buffers.py:143-155, :312-323)."""
        with th.cuda.device(self.device):
            hip_ops.vecnorm_apply(self.cfg, self._state, obs, next_obs, rewards)

    # ---- reference API (NumPy in / out) ---------------------------------------------------------------------------
    def reset(self) -> np.ndarray:
        return self.reset_device().cpu().numpy()

    def step_async(self, actions) -> None:
        self.venv.step_async(actions)

    def step_wait(self):
        obs, rewards, dones, infos = self.venv.step_wait()
        self.old_obs, self.old_reward = obs, rewards
        n_obs = self.after_device_step().cpu().numpy()
        n_rew = self.norm_rew_dev.cpu().numpy()
        for i, d in enumerate(dones):
            if d and "terminal_observation" in infos[i]:
                infos[i]["terminal_observation"] = self.normalize_obs(infos[i]["terminal_observation"])
        return n_obs, n_rew, dones, infos

    def _moments(self):
        return self.obs_rms.mean, self.obs_rms.var, float(self.ret_rms.var)

    def normalize_obs(self, obs):
        """reference: vec_normalize.py:225-241 (does not update the statistics)"""
        if isinstance(obs, th.Tensor):
            out = obs.to(self.device, th.float32).reshape(-1, self.venv.obs_dim).contiguous().clone()
            with th.cuda.device(self.device):
                hip_ops.vecnorm_apply(self.cfg, self._state, out, None, None)
            return out.reshape(obs.shape)
        if not self.norm_obs:
            return np.array(obs, copy=True)
        mean, var, _ = self._moments()
        return np.clip((obs - mean) / np.sqrt(var + self.epsilon), -self.clip_obs, self.clip_obs).astype(np.float32)

    def normalize_reward(self, reward):
        """reference: vec_normalize.py:243-252"""
        reward = np.asarray(reward)
        if self.norm_reward:
            reward = np.clip(reward / np.sqrt(self._moments()[2] + self.epsilon), -self.clip_reward, self.clip_reward)
        return reward.astype(np.float32)

    def unnormalize_obs(self, obs):
        """reference: vec_normalize.py:216-223, :254-266"""
        if not self.norm_obs:
            return np.array(obs, copy=True)
        mean, var, _ = self._moments()
        return (np.asarray(obs) * np.sqrt(var + self.epsilon)) + mean

    def unnormalize_reward(self, reward):
        """reference: vec_normalize.py:268-271"""
        return reward * np.sqrt(self._moments()[2] + self.epsilon) if self.norm_reward else reward

    def get_original_obs(self) -> np.ndarray:
        return self.venv.obs.cpu().numpy()

    def get_original_reward(self) -> np.ndarray:
        return self.venv._rew.cpu().numpy()

    # ---- persistence (vec_normalize.py:128-172, :309-345) ---------------------------------------------------------
    def __getstate__(self) -> dict:
        return dict(training=self.training, norm_obs=self.norm_obs, norm_reward=self.norm_reward, clip_obs=self.clip_obs,
                    clip_reward=self.clip_reward, gamma=self.gamma, epsilon=self.epsilon, observation_space=self.observation_space,
                    action_space=self.action_space, num_envs=self.num_envs, statistics=self._state.cpu().numpy())

    def __setstate__(self, st: dict) -> None:
        self.__dict__.update({k: v for k, v in st.items() if k != "statistics"})
        self._saved_statistics, self.venv = st["statistics"], None  # set_venv() finishes the job

    def set_venv(self, venv: CSTRVecEnv) -> None:
        if self.__dict__.get("venv") is not None:
            raise ValueError("Trying to set venv of already initialized VecNormalize wrapper.")
        saved = self.__dict__.pop("_saved_statistics")
        if tuple(venv.observation_space.shape) != tuple(self.observation_space.shape):
            raise ValueError("spaces must have the same shape")
        kw = {k: self.__dict__[k] for k in ("training", "norm_obs", "norm_reward", "clip_obs", "clip_reward", "gamma", "epsilon")}
        VecNormalize.__init__(self, venv, **kw)
        self._state.copy_(th.from_numpy(np.asarray(saved, np.float64)))

    @staticmethod
    def load(load_path: str, venv: CSTRVecEnv) -> "VecNormalize":
        with open(load_path, "rb") as f:
            vn = pickle.load(f)
        vn.set_venv(venv)
        return vn

    def save(self, save_path: str) -> None:
        with open(save_path, "wb") as f:
            pickle.dump(self, f)


def unwrap_vec_normalize(env) -> Optional[VecNormalize]:
    """reference: core/common/vec_env/__init__.py:42-50"""
    return env if isinstance(env, VecNormalize) else None
