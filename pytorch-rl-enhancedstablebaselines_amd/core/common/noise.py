"""Exploration noise (reference: core/common/noise.py:9-174).

The reference draws Gaussian noise with `np.random.normal` on the process-global legacy stream (:44-45) -- the same
stream ReplayBuffer.sample draws its indices from. On the device env's fast path `NormalActionNoise` is therefore
replaced by `LegacyStreamNormalActionNoise`, which draws all n_envs x action_dim deviates in ONE kernel from the HBM
image of that stream (cstr_mt19937_normal_f32): the noise values and the replay index stream that follows stay
bit-faithful to a seeded reference run; `OrnsteinUhlenbeckActionNoise` likewise becomes
`LegacyStreamOUActionNoise` (float64 draws from the same stream). `DeviceNormalActionNoise` is the faster, statistically
equivalent alternative (torch's device generator, not bit-faithful). The plain host classes keep the reference's
NumPy code for generic (non-device) envs; their host draws do NOT advance the device stream."""
import copy
from typing import Iterable, Optional

import numpy as np


class ActionNoise:
    def reset(self) -> None:
        pass

    def __call__(self) -> np.ndarray:
        raise NotImplementedError


class NormalActionNoise(ActionNoise):
    def __init__(self, mean: np.ndarray, sigma: np.ndarray, dtype=np.float32):
        self._mu, self._sigma, self._dtype = mean, sigma, dtype

    def __call__(self) -> np.ndarray:
        return np.random.normal(self._mu, self._sigma).astype(self._dtype)

    def __repr__(self) -> str:
        return f"NormalActionNoise(mu={self._mu}, sigma={self._sigma})"


class OrnsteinUhlenbeckActionNoise(ActionNoise):
    def __init__(self, mean, sigma, theta: float = 0.15, dt: float = 1e-2, initial_noise: Optional[np.ndarray] = None,
                 dtype=np.float32):
        self._theta, self._mu, self._sigma, self._dt, self._dtype = theta, mean, sigma, dt, dtype
        self.initial_noise = initial_noise
        self.noise_prev = np.zeros_like(self._mu)
        self.reset()

    def __call__(self) -> np.ndarray:
        noise = (self.noise_prev + self._theta * (self._mu - self.noise_prev) * self._dt
                 + self._sigma * np.sqrt(self._dt) * np.random.normal(size=self._mu.shape))
        self.noise_prev = noise
        return noise.astype(self._dtype)

    def reset(self) -> None:
        self.noise_prev = self.initial_noise if self.initial_noise is not None else np.zeros_like(self._mu)


class VectorizedActionNoise(ActionNoise):
    """reference: noise.py:108-174 -- one independent noise process per env"""

    def __init__(self, base_noise: ActionNoise, n_envs: int):
        self.n_envs = int(n_envs)
        assert self.n_envs > 0
        self.base_noise = base_noise
        self.noises = [copy.deepcopy(base_noise) for _ in range(self.n_envs)]

    def reset(self, indices: Optional[Iterable[int]] = None) -> None:
        for i in (range(len(self.noises)) if indices is None else indices):
            self.noises[i].reset()

    def __call__(self) -> np.ndarray:
        return np.stack([noise() for noise in self.noises])


class LegacyStreamNormalActionNoise(ActionNoise):
    """`VectorizedActionNoise(NormalActionNoise(mean, sigma), n_envs)()` (reference: noise.py:44-45, :141-142) on the
    GPU: n_envs consecutive `np.random.normal(mean, sigma).astype(float32)` draws from the legacy MT19937 stream that
    `stream_fn()` returns (the replay sampler's), in numpy's order. Graph-capturable: all state lives in HBM."""

    def __init__(self, mean, sigma, n_envs: int, device, stream_fn):
        import torch as th

        self._mu = np.asarray(mean, np.float64).reshape(-1)
        self._sigma = np.broadcast_to(np.asarray(sigma, np.float64).reshape(-1), self._mu.shape).copy()
        self.n_envs, self.device, self._stream_fn = int(n_envs), th.device(device), stream_fn
        self._out = th.empty(self.n_envs, self._mu.size, dtype=th.float32, device=self.device)

    def __call__(self):
        from core.common import hip_ops

        hip_ops.mt19937_normal(self._stream_fn(), self._mu, self._sigma, self._out)
        return self._out

    def reset(self, indices: Optional[Iterable[int]] = None) -> None:
        pass  # memoryless

    def __repr__(self) -> str:
        return f"LegacyStreamNormalActionNoise(mu={self._mu.tolist()}, sigma={self._sigma.tolist()}, n_envs={self.n_envs})"


class LegacyStreamOUActionNoise(ActionNoise):
    """`VectorizedActionNoise(OrnsteinUhlenbeckActionNoise(...), n_envs)()` (reference: noise.py:48-106, :141-142) on the GPU:
    the n_envs x action_dim standard normals come from the legacy MT19937 stream `stream_fn()` returns (the replay
    sampler's) in numpy's draw order and in float64, the recursion runs in float64 with the reference's operation order,
    the result is cast to float32 like `.astype(self._dtype)`. `reset_done(done)` = `reset(indices)` for finished envs."""

    def __init__(self, mean, sigma, theta: float, dt: float, initial_noise, n_envs: int, device, stream_fn):
        import torch as th

        self._mu_np = np.asarray(mean, np.float64).reshape(-1)
        a = self._mu_np.size
        self._sigma_np = np.broadcast_to(np.asarray(sigma, np.float64).reshape(-1), (a,)).copy()
        self._theta, self._dt = float(theta), float(dt)
        self.n_envs, self.device, self._stream_fn = int(n_envs), th.device(device), stream_fn
        dev = self.device
        self._mu = th.as_tensor(self._mu_np, device=dev).reshape(1, a)
        self._s = th.as_tensor(self._sigma_np * np.sqrt(self._dt), device=dev).reshape(1, a)  # sigma * np.sqrt(dt) (:87)
        init = np.zeros(a) if initial_noise is None else np.asarray(initial_noise, np.float64).reshape(-1)
        self._init = th.as_tensor(init, device=dev).reshape(1, a).expand(self.n_envs, a).contiguous()
        self.noise_prev = self._init.clone()
        self._z = th.empty(self.n_envs, a, dtype=th.float64, device=dev)
        self._zero, self._one = [0.0] * a, [1.0] * a

    def __call__(self):
        from core.common import hip_ops

        hip_ops.mt19937_normal(self._stream_fn(), self._zero, self._one, self._z)  # np.random.normal(size=mu.shape), per env
        p = self.noise_prev
        noise = p + self._theta * (self._mu - p) * self._dt + self._s * self._z  # the reference's operation order (:84-88)
        self.noise_prev.copy_(noise)
        return noise.float()  # .astype(np.float32)

    def reset(self, indices: Optional[Iterable[int]] = None) -> None:
        if indices is None:
            self.noise_prev.copy_(self._init)
        else:
            idx = list(indices)
            if idx:
                self.noise_prev[idx] = self._init[idx]

    def reset_done(self, done) -> None:
        """reset(indices) for the envs whose episode just ended (off_policy_algorithm.py:596-599), without a host sync"""
        import torch as th

        self.noise_prev.copy_(th.where(done.reshape(-1, 1) > 0, self._init, self.noise_prev))

    def __repr__(self) -> str:
        return f"LegacyStreamOUActionNoise(mu={self._mu_np.tolist()}, sigma={self._sigma_np.tolist()}, n_envs={self.n_envs})"


class DeviceNormalActionNoise(ActionNoise):
    """Gaussian exploration noise drawn on the GPU for all envs at once: N(mean, sigma) of shape [n_envs, action_dim],
    graph-capturable. What `VectorizedActionNoise(NormalActionNoise(...), n_envs)` (reference: noise.py:29-45, :108-174)
    computes, but from torch's device generator instead of n_envs sequential `np.random.normal` calls on the legacy
    global stream -- statistically identical, NOT bit-identical."""

    def __init__(self, mean, sigma, n_envs: int, device):
        import torch as th

        self._mu = th.as_tensor(np.asarray(mean, np.float32), device=device).reshape(1, -1)
        self._sigma = th.as_tensor(np.asarray(sigma, np.float32), device=device).reshape(1, -1)
        self.n_envs, self.device = n_envs, device

    def __call__(self):
        import torch as th

        return self._mu + self._sigma * th.randn(self.n_envs, self._mu.shape[1], device=self.device)

    def reset(self, indices: Optional[Iterable[int]] = None) -> None:
        pass  # memoryless

    def __repr__(self) -> str:
        return f"DeviceNormalActionNoise(mu={self._mu.flatten().tolist()}, sigma={self._sigma.flatten().tolist()}, n_envs={self.n_envs})"
