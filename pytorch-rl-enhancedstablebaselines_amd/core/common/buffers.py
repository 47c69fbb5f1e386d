"""`ReplayBuffer` with the reference's constructor, attributes and `add`/`sample` semantics
(reference: core/common/buffers.py:27-340), stored in HBM and served by HIP kernels:

  * field arrays `observations / next_observations / actions / rewards / dones / timeouts` keep the
    reference's shapes `[rows, n_envs, ...]` and are device tensors (zero-copy views of the ring);
  * `add` is one launch (row write at the device-resident position);
  * `sample` draws `np.random.randint` indices from the device image of NumPy's legacy global MT19937
    stream -- bit-identical to the reference's `batch_inds` / `env_indices` -- and gathers the five
    fields in the same launch.
"""
from typing import Any, Optional, Union

import numpy as np
import torch as th

from core.common import hip_ops, legacy_rng
from core.common.spaces import as_box, get_action_dim, get_obs_shape
from core.common.type_aliases import ReplayBufferSamples
from core.common.utils import get_device


class BaseBuffer:
    """reference: core/common/buffers.py:27-156 (the parts the off-policy path uses)"""

    def __init__(self, buffer_size: int, observation_space, action_space, device: Union[th.device, str] = "auto",
                 n_envs: int = 1):
        self.buffer_size = buffer_size
        self.observation_space = as_box(observation_space)
        self.action_space = as_box(action_space)
        self.obs_shape = get_obs_shape(self.observation_space)
        self.action_dim = get_action_dim(self.action_space)
        self.device = get_device(device)
        self.n_envs = n_envs
        self._adds = 0  # host mirror of the device ring position (every add goes through this object)

    @property
    def pos(self) -> int:
        return self._adds % self.buffer_size

    @property
    def full(self) -> bool:
        return self._adds >= self.buffer_size

    def size(self) -> int:
        """reference: buffers.py:69-75"""
        return self.buffer_size if self.full else self.pos

    def reset(self) -> None:
        raise NotImplementedError

    def to_torch(self, array, copy: bool = True) -> th.Tensor:
        """reference: buffers.py:128-140"""
        if isinstance(array, th.Tensor):
            return array.to(self.device, th.float32)
        return th.tensor(array, device=self.device, dtype=th.float32) if copy else th.as_tensor(array, device=self.device, dtype=th.float32)


class ReplayBuffer(BaseBuffer):
    """reference: core/common/buffers.py:158-340.

    :param buffer_size: max number of transitions; the ring has `max(buffer_size // n_envs, 1)` rows (:198)
    :param sampler_stream: optional private MT19937 state (uint32[628] as int32, HBM); default = the device image of
        NumPy's global legacy stream (`np.random.randint`, :113/:309)
    """

    def __init__(self, buffer_size: int, observation_space, action_space, device: Union[th.device, str] = "auto",
                 n_envs: int = 1, optimize_memory_usage: bool = False, handle_timeout_termination: bool = True,
                 sampler_stream: Optional[th.Tensor] = None):
        super().__init__(buffer_size, observation_space, action_space, device, n_envs=n_envs)
        self.buffer_size = max(buffer_size // n_envs, 1)  # :198
        if optimize_memory_usage and handle_timeout_termination:
            raise ValueError("ReplayBuffer does not support optimize_memory_usage = True "
                             "and handle_timeout_termination = True simultaneously.")  # :206-210
        if optimize_memory_usage:
            raise NotImplementedError("optimize_memory_usage=True is not built (incompatible with timeout handling, "
                                      "which the CSTR path needs; SURVEY a-9)")
        self.optimize_memory_usage = optimize_memory_usage
        self.handle_timeout_termination = handle_timeout_termination
        if len(self.obs_shape) != 1:
            raise ValueError(f"ReplayBuffer supports flat Box observations, got shape {self.obs_shape}")
        with th.cuda.device(self.device):
            self.ring = hip_ops.DeviceRing(self.buffer_size, n_envs, self.obs_shape[0], self.action_dim, self.device)
        r = self.ring
        self.observations, self.next_observations, self.actions = r.observations, r.next_observations, r.actions
        self.rewards, self.dones, self.timeouts = r.rewards, r.dones, r.timeouts
        self._stream = sampler_stream
        self.normalizer = None  # a VecNormalize whose statistics normalise every sampled batch (buffers.py:143-155)
        self._predrawn = None  # (sample_idx, rng_advance) of a rollout launch whose indices the next packed sample gathers

    # ---- pickling (save_replay_buffer / load_replay_buffer, off_policy_algorithm.py:214-254) -------------
    _FIELDS = ("observations", "next_observations", "actions", "rewards", "dones", "timeouts")

    def __getstate__(self) -> dict:
        """The pickled form carries the reference's attributes (buffers.py:212-234) as host NumPy arrays with the
        reference's shapes [rows, n_envs, ...], `pos` and `full`; the HBM ring is rebuilt on load."""
        st = {k: getattr(self, k) for k in ("buffer_size", "observation_space", "action_space", "obs_shape", "action_dim", "n_envs",
                                            "optimize_memory_usage", "handle_timeout_termination")}
        st.update({k: getattr(self, k).cpu().numpy() for k in self._FIELDS})
        st.update(pos=self.pos, full=self.full, device=str(self.device), adds=self._adds,
                  sampler_stream=None if self._stream is None else self._stream.cpu().numpy())
        return st

    def __setstate__(self, st: dict) -> None:
        for k in ("buffer_size", "observation_space", "action_space", "obs_shape", "action_dim", "n_envs", "optimize_memory_usage",
                  "handle_timeout_termination"):
            setattr(self, k, st[k])
        self.device = get_device(st.get("device", "auto"))
        with th.cuda.device(self.device):
            self.ring = hip_ops.DeviceRing(self.buffer_size, self.n_envs, self.obs_shape[0], self.action_dim, self.device)
        r = self.ring
        self.observations, self.next_observations, self.actions = r.observations, r.next_observations, r.actions
        self.rewards, self.dones, self.timeouts = r.rewards, r.dones, r.timeouts
        for k in self._FIELDS:
            getattr(self, k).copy_(th.from_numpy(np.ascontiguousarray(st[k], dtype=np.float32)).reshape(getattr(self, k).shape))
        # a pickle written by the reference has pos/full only; `adds` keeps the device's running add counter exact
        self._adds = int(st.get("adds", st["pos"] + (self.buffer_size if st["full"] else 0)))
        r.ctl.copy_(th.tensor([int(st["pos"]), int(bool(st["full"])), 0, self._adds], dtype=th.int64))
        ss = st.get("sampler_stream")
        self._stream = None if ss is None else th.from_numpy(np.ascontiguousarray(ss)).to(self.device)
        self.normalizer = None
        self._predrawn = None

    def to(self, device) -> "ReplayBuffer":
        """Move the ring to another GPU (load_replay_buffer: 'update saved replay buffer device', :252-253)."""
        device = get_device(device)
        if device != self.device:
            st = self.__getstate__()
            st["device"] = str(device)
            self.__setstate__(st)
        return self

    # ---- sampler stream ---------------------------------------------------------------------------------
    @property
    def sampler_stream(self) -> th.Tensor:
        return self._stream if self._stream is not None else legacy_rng.global_stream(self.device)

    def seed_sampler(self, seed: int) -> None:
        """Give this buffer a private stream = np.random.RandomState(seed) (data-parallel shards, tests)."""
        self._stream = legacy_rng.new_stream(seed, self.device)

    # ---- add --------------------------------------------------------------------------------------------
    def _dev(self, x, shape) -> th.Tensor:
        if isinstance(x, th.Tensor):
            t = x.to(self.device, th.float32)
        else:
            t = th.as_tensor(np.ascontiguousarray(np.asarray(x), dtype=np.float32)).to(self.device)
        return t.reshape(shape).contiguous()

    def add(self, obs, next_obs, action, reward, done, infos: Optional[list]) -> None:
        """reference: buffers.py:247-283. Accepts NumPy (compatibility) or device tensors. `infos` may be the
        reference's list of dicts or a device/NumPy vector of timeout flags."""
        n = self.n_envs
        self._no_predrawn("ReplayBuffer.add")
        if infos is None or not self.handle_timeout_termination:
            timeout = th.zeros(n, dtype=th.float32, device=self.device)
        elif isinstance(infos, (th.Tensor, np.ndarray)):
            timeout = self._dev(infos, (n,))
        else:
            timeout = self._dev(np.array([info.get("TimeLimit.truncated", False) for info in infos]), (n,))  # :277-278
        with th.cuda.device(self.device):
            hip_ops.replay_add(self.ring, self._dev(obs, (n, *self.obs_shape)), self._dev(next_obs, (n, *self.obs_shape)),
                               self._dev(action, (n, self.action_dim)), self._dev(reward, (n,)), self._dev(done, (n,)),
                               timeout)
        self._adds += 1

    def note_fused_add(self) -> None:
        """The fused collect kernel wrote a row and advanced the device position; keep the host mirror in step."""
        self._adds += 1

    # ---- rollout launch that also draws the next sample's indices (hip_ops.rollout_step) -----------------------------
    def predraw_indices(self, batch_size: int) -> th.Tensor:
        """Static int32 [2, batch] buffer the rollout launch writes (batch_inds, env_indices) into."""
        buf = getattr(self, "_predraw_buf", None)
        if buf is None or buf.shape[1] != batch_size or buf.device != self.ring.ctl.device:  # to(device) / unpickling move the ring
            buf = self._predraw_buf = th.zeros(2, batch_size, dtype=th.int32, device=self.device)
        return buf

    def note_predrawn(self, sample_idx: th.Tensor, rng_advance=None) -> None:
        """A rollout launch wrote the ring row at the device position WITHOUT advancing it and drew `sample_idx` for the sample
        behind it: the next `sample_packed_into` must gather by these indices and advance the control words. Anything else that
        touches the ring first is a bug in the caller (raised, not papered over)."""
        if self._predrawn is not None:
            raise RuntimeError("a pre-drawn sample is already pending")
        self._predrawn = (sample_idx, rng_advance)

    def take_predrawn(self, pb: "PackedBatch"):
        """Hand the pending pre-drawn sample to a consumer that gathers the rows itself (hip_ops.linear_act_fwd_gather: the first layer
        behind the sample) and performs the control-word updates: returns (ring, sample_idx, rng_advance, pb), or None if no
        pre-drawn sample of this batch size is pending."""
        if self._predrawn is None or self._predrawn[0].shape[1] != pb.x_data.shape[0] or self.normalizer is not None:
            return None
        (idx, rng_advance), self._predrawn = self._predrawn, None
        return self.ring, idx, rng_advance, pb

    def _no_predrawn(self, what: str) -> None:
        if self._predrawn is not None:
            raise RuntimeError(f"{what}: a rollout launch left its ring advance to the next packed sample, which has not run yet")

    def reset(self) -> None:
        self._adds = 0
        self.ring.ctl.zero_()

    # ---- sample -----------------------------------------------------------------------------------------
    def alloc_batch(self, batch_size: int):
        d, a, dev = self.obs_shape[0], self.action_dim, self.device
        e = lambda *s: th.empty(*s, dtype=th.float32, device=dev)  # noqa: E731
        return ReplayBufferSamples(e(batch_size, d), e(batch_size, a), e(batch_size, d), e(batch_size, 1), e(batch_size, 1))

    def sample_into(self, out: ReplayBufferSamples, row_idx=None, env_idx=None, env=None) -> ReplayBufferSamples:
        """`sample` into caller-owned (static, graph-capturable) tensors."""
        if self.size() == 0:
            raise ValueError("high <= 0")  # what np.random.randint(0, 0) raises in the reference (:113)
        self._no_predrawn("ReplayBuffer.sample")
        with th.cuda.device(self.device):
            hip_ops.replay_sample(self.ring, self.sampler_stream, out.observations.shape[0], out.observations, out.actions,
                                  out.next_observations, out.dones, out.rewards, row_idx, env_idx)
        env = self.normalizer if env is None else env
        if env is not None:  # _normalize_obs / _normalize_reward with the CURRENT statistics (:143-155, :312-323)
            env.normalize_batch_(out.observations, out.next_observations, out.rewards)
        return out

    # ---- packed batches: sampled rows land directly in the critics' input buffers -------------------------------
    def alloc_packed_batch(self, batch_size: int, with_pi: bool = True) -> "PackedBatch":
        return PackedBatch(batch_size, self.obs_shape[0], self.action_dim, self.device, with_pi)

    def sample_packed_into(self, pb: "PackedBatch", row_idx=None, env_idx=None) -> "PackedBatch":
        """`sample` + the critics' torch.cat([obs, actions], 1) (core/common/policies.py:975-981) in one launch. Not
        available with a VecNormalize normaliser (its kernel works on contiguous observation batches)."""
        if self.size() == 0:
            raise ValueError("high <= 0")
        if self.normalizer is not None:
            raise ValueError("packed batches do not go through VecNormalize; use sample_into")
        if self._predrawn is not None:  # indices drawn by the rollout launch in front of this call: gather + control-word updates
            (idx, rng_advance), self._predrawn = self._predrawn, None
            if idx.shape[1] != pb.x_data.shape[0]:
                raise RuntimeError(f"pre-drawn sample of {idx.shape[1]} rows, batch of {pb.x_data.shape[0]}")
            with th.cuda.device(self.device):
                hip_ops.replay_gather_packed(self.ring, idx, idx.shape[1], pb.x_data, pb.x_next, pb.x_pi, pb.samples.dones,
                                             pb.samples.rewards, row_idx, env_idx, advance_ring=True, rng_advance=rng_advance)
            return pb
        with th.cuda.device(self.device):
            hip_ops.replay_sample_packed(self.ring, self.sampler_stream, pb.x_data.shape[0], pb.x_data, pb.x_next, pb.x_pi,
                                         pb.samples.dones, pb.samples.rewards, row_idx, env_idx)
        return pb

    def sample(self, batch_size: int, env: Any = None) -> ReplayBufferSamples:
        """reference: buffers.py:106-115, :285-325. Returns fresh device tensors
        (observations, actions, next_observations, dones, rewards)."""
        return self.sample_into(self.alloc_batch(batch_size), env=env)

    def sample_with_indices(self, batch_size: int):
        bi = th.empty(batch_size, dtype=th.int64, device=self.device)
        ei = th.empty(batch_size, dtype=th.int64, device=self.device)
        return self.sample_into(self.alloc_batch(batch_size), bi, ei), bi, ei


class PackedBatch:
    """Static critic-input buffers of one sampled batch, W = obs_dim + act_dim columns each:
    x_data = (obs | act), x_next = (next_obs | next action, written by the actor head), x_pi = (obs | pi(obs), likewise).
    `samples` exposes the usual ReplayBufferSamples fields as (row-strided) views of them."""

    def __init__(self, batch_size: int, obs_dim: int, act_dim: int, device, with_pi: bool = True):
        e = lambda *s: th.zeros(*s, dtype=th.float32, device=device)  # noqa: E731
        w = obs_dim + act_dim
        self.obs_dim, self.act_dim = obs_dim, act_dim
        self.x_data = e(batch_size, w)
        # x_pi and x_next are the two halves of ONE [2B, W] buffer: SAC evaluates pi(obs) and pi(next_obs) in one 2B-row actor
        # pass (fused._ActorPairFn) that reads the observation columns and writes the action columns with one row stride
        self.x_pn = e(2 * batch_size, w) if with_pi else None
        self.x_next = self.x_pn[batch_size:] if with_pi else e(batch_size, w)
        self.x_pi = self.x_pn[:batch_size] if with_pi else None
        self.samples = ReplayBufferSamples(self.x_data[:, :obs_dim], self.x_data[:, obs_dim:], self.x_next[:, :obs_dim],
                                           e(batch_size, 1), e(batch_size, 1))
