"""Torch-tensor front end of the C ABI (include/cstr_rl_hip.h).

Every wrapper validates shape / dtype / device / contiguity on the host before the launch (a
mis-shaped operand must never reach a hand-written kernel), passes raw device pointers and the
current HIP stream, and raises on a non-zero return code. Nothing here computes on the CPU.
"""
import ctypes as C
from typing import Optional

import torch as th

from core import _native as nv
from core._native import INTEGRATORS, check, ptr, stream_ptr


def _chk(t: th.Tensor, name: str, shape, dtype) -> th.Tensor:
    if not isinstance(t, th.Tensor):
        raise TypeError(f"{name}: expected a torch.Tensor, got {type(t)}")
    if not t.is_cuda:
        raise ValueError(f"{name}: must live in HBM (device tensor); got device {t.device}. No CPU fallback exists.")
    if t.device.index != th.cuda.current_device():  # a pointer from another GPU must never reach a launch on this one
        raise ValueError(f"{name}: lives on {t.device}, the current device is cuda:{th.cuda.current_device()}")
    if t.dtype != dtype:
        raise ValueError(f"{name}: dtype {t.dtype}, expected {dtype}")
    if tuple(t.shape) != tuple(shape):
        raise ValueError(f"{name}: shape {tuple(t.shape)}, expected {tuple(shape)}")
    if not t.is_contiguous():
        raise ValueError(f"{name}: must be contiguous")
    return t


LAYOUTS = ((4, 2), (8, 2), (8, 4))


def _opt(t: Optional[th.Tensor], name, shape, dtype):
    return None if t is None else _chk(t, name, shape, dtype)


class DeviceRing:
    """cstr_ring_t + ring_ctl over torch tensors (HBM). Field arrays have the reference's shapes
    (core/common/buffers.py:212-234): [rows, n_envs, D] / [rows, n_envs, A] / [rows, n_envs]."""

    def __init__(self, rows: int, n_envs: int, obs_dim: int, act_dim: int, device):
        if (obs_dim, act_dim) not in LAYOUTS:
            raise ValueError(f"DeviceRing supports (obs_dim, act_dim) in (4,2), (8,2), (8,4) (CSTR layouts), got {obs_dim}/{act_dim}")
        z = lambda *s: th.zeros(*s, dtype=th.float32, device=device)  # noqa: E731
        self.observations, self.next_observations = z(rows, n_envs, obs_dim), z(rows, n_envs, obs_dim)
        self.actions = z(rows, n_envs, act_dim)
        self.rewards, self.dones, self.timeouts = z(rows, n_envs), z(rows, n_envs), z(rows, n_envs)
        self.ctl = th.zeros(nv.RING_CTL_WORDS, dtype=th.int64, device=device)  # {pos, full, ticket, adds}
        self.rows, self.n_envs, self.obs_dim, self.act_dim = rows, n_envs, obs_dim, act_dim
        self.c = nv.Ring(self.observations.data_ptr(), self.next_observations.data_ptr(), self.actions.data_ptr(),
                         self.rewards.data_ptr(), self.dones.data_ptr(), self.timeouts.data_ptr(), rows, n_envs,
                         obs_dim, act_dim)


def vec_step(coef, integrator: str, obs, act, step_count, reset_obs, next_obs, obs_after, reward, done, timeout):
    n, d = obs.shape
    a = act.shape[-1]
    if (d, a) not in LAYOUTS:
        raise ValueError(f"(obs_dim, act_dim) = {(d, a)} is not a CSTR layout {LAYOUTS}")
    _chk(obs, "obs", (n, d), th.float32), _chk(act, "act", (n, a), th.float32)
    _chk(step_count, "step_count", (n,), th.int32), _chk(reset_obs, "reset_obs", (n, d), th.float32)
    _chk(next_obs, "next_obs", (n, d), th.float32), _chk(obs_after, "obs_after", (n, d), th.float32)
    for t, nm in ((reward, "reward"), (done, "done"), (timeout, "timeout")):
        _chk(t, nm, (n,), th.float32)
    check(nv.lib().cstr_vec_step_f32(C.byref(coef), C.c_int(INTEGRATORS[integrator]), C.c_int(d), C.c_int(a), ptr(obs), ptr(act),
                                     ptr(step_count), ptr(reset_obs), ptr(next_obs), ptr(obs_after), ptr(reward),
                                     ptr(done), ptr(timeout), C.c_int64(n), stream_ptr()), "cstr_vec_step_f32")


def reset_draw(pcg_state, mask, obs_out, act_dim: int = 2, static_init=None):
    n, d = obs_out.shape
    if (d, act_dim) not in LAYOUTS:
        raise ValueError(f"(obs_dim, act_dim) = {(d, act_dim)} is not a CSTR layout {LAYOUTS}")
    _chk(pcg_state, "pcg_state", (n, nv.PCG_STATE_WORDS), th.int64), _chk(obs_out, "obs_out", (n, d), th.float32)
    _opt(mask, "mask", (n,), th.uint8)
    _opt(static_init, "static_init", (n, 8 if act_dim == 4 else 4), th.float64)
    check(nv.lib().cstr_reset_draw_f32(ptr(pcg_state), ptr(mask), ptr(static_init), C.c_int(d), C.c_int(act_dim), ptr(obs_out),
                                       C.c_int64(n), stream_ptr()), "cstr_reset_draw_f32")


def replay_add(ring: DeviceRing, obs, next_obs, act, rew, done, timeout):
    n, d, a = ring.n_envs, ring.obs_dim, ring.act_dim
    _chk(obs, "obs", (n, d), th.float32), _chk(next_obs, "next_obs", (n, d), th.float32), _chk(act, "act", (n, a), th.float32)
    for t, nm in ((rew, "rew"), (done, "done"), (timeout, "timeout")):
        _chk(t, nm, (n,), th.float32)
    check(nv.lib().cstr_replay_add_f32(C.byref(ring.c), ptr(ring.ctl), ptr(obs), ptr(next_obs), ptr(act), ptr(rew),
                                       ptr(done), ptr(timeout), stream_ptr()), "cstr_replay_add_f32")


def collect_step(coef, integrator: str, ring: DeviceRing, env_obs, step_count, policy_out, squashed, act_low,
                 act_high, noise=None, reset_obs=None, pcg_state=None, reward_out=None, done_out=None, ep_return=None,
                 ep_stats=None, static_init=None, rng_advance=None):
    """`rng_advance` = (rng_ctl, count) or None: the Philox control block of the policy launch that produced `policy_out` with
    `defer_rng_advance=True`; this launch's last workgroup advances its offset by `count` (cstr_collect_step_rng_f32)."""
    n, d, a = ring.n_envs, ring.obs_dim, ring.act_dim
    _chk(env_obs, "env_obs", (n, d), th.float32), _chk(step_count, "step_count", (n,), th.int32)
    _chk(policy_out, "policy_out", (n, a), th.float32)
    _opt(noise, "noise", (n, a), th.float32), _opt(reset_obs, "reset_obs", (n, d), th.float32)
    _opt(pcg_state, "pcg_state", (n, nv.PCG_STATE_WORDS), th.int64)
    _opt(static_init, "static_init", (n, 8 if a == 4 else 4), th.float64)
    if static_init is not None and pcg_state is None:
        raise ValueError("static_init (init_mode='static') needs the per-env pcg_state reset source")
    _opt(reward_out, "reward_out", (n,), th.float32), _opt(done_out, "done_out", (n,), th.float32)
    _opt(ep_return, "ep_return", (n,), th.float32), _opt(ep_stats, "ep_stats", (4,), th.float64)
    if (ep_return is None) != (ep_stats is None):
        raise ValueError("ep_return and ep_stats go together")
    if (reset_obs is None) == (pcg_state is None):
        raise ValueError("collect_step needs exactly one reset source: reset_obs or pcg_state")
    if len(act_low) != a or len(act_high) != a:
        raise ValueError(f"act_low/act_high need {a} entries")
    lo = (C.c_float * a)(*[float(v) for v in act_low])
    hi = (C.c_float * a)(*[float(v) for v in act_high])
    rng_ctl, rng_count = rng_advance if rng_advance is not None else (None, 0)
    _opt(rng_ctl, "rng_ctl", (nv.RNG_CTL_WORDS,), th.int64)
    check(nv.lib().cstr_collect_step_rng_f32(C.byref(coef), C.c_int(INTEGRATORS[integrator]), C.byref(ring.c), ptr(ring.ctl),
                                             ptr(env_obs), ptr(step_count), ptr(policy_out), C.c_int(int(squashed)), lo, hi,
                                             ptr(noise), ptr(reset_obs), ptr(pcg_state), ptr(static_init), ptr(reward_out), ptr(done_out),
                                             ptr(ep_return), ptr(ep_stats), ptr(rng_ctl), C.c_uint64(int(rng_count)), stream_ptr()),
          "cstr_collect_step_rng_f32")


def mt19937_seed(mt_state, seed: int):
    _chk(mt_state, "mt_state", (nv.MT_STATE_WORDS,), th.int32)
    check(nv.lib().cstr_mt19937_seed(ptr(mt_state), C.c_uint32(seed & 0xFFFFFFFF), stream_ptr()), "cstr_mt19937_seed")


def mt19937_normal(mt_state, loc, scale, out):
    """`out.shape[0]` consecutive np.random.normal(loc, scale).astype(float32) from the device legacy stream
    (reference: core/common/noise.py:44-45, :141-142); out [n, len(loc)] f32."""
    _chk(mt_state, "mt_state", (nv.MT_STATE_WORDS,), th.int32)
    loc, scale = [float(v) for v in loc], [float(v) for v in scale]
    period = len(loc)
    if len(scale) != period or not 1 <= period <= nv.MAX_NOISE_PERIOD:
        raise ValueError(f"loc / scale must have the same length in [1, {nv.MAX_NOISE_PERIOD}]")
    if any(not v >= 0.0 for v in scale):
        raise ValueError("scale < 0")  # numpy's message
    if out.dtype not in (th.float32, th.float64):
        raise ValueError(f"out: dtype {out.dtype}, expected float32 or float64")
    _chk(out, "out", (out.shape[0], period), out.dtype)
    fn = nv.lib().cstr_mt19937_normal_f32 if out.dtype == th.float32 else nv.lib().cstr_mt19937_normal_f64
    check(fn(ptr(mt_state), (C.c_double * period)(*loc), (C.c_double * period)(*scale), C.c_int32(period), ptr(out),
             C.c_int64(out.numel()), stream_ptr()), "cstr_mt19937_normal")


def replay_sample(ring: DeviceRing, mt_state, batch: int, out_obs, out_act, out_next_obs, out_done, out_rew,
                  out_row_idx=None, out_env_idx=None):
    d, a = ring.obs_dim, ring.act_dim
    _chk(mt_state, "mt_state", (nv.MT_STATE_WORDS,), th.int32)
    _chk(out_obs, "out_obs", (batch, d), th.float32), _chk(out_next_obs, "out_next_obs", (batch, d), th.float32)
    _chk(out_act, "out_act", (batch, a), th.float32)
    _chk(out_done, "out_done", (batch, 1), th.float32), _chk(out_rew, "out_rew", (batch, 1), th.float32)
    _opt(out_row_idx, "out_row_idx", (batch,), th.int64), _opt(out_env_idx, "out_env_idx", (batch,), th.int64)
    if not 0 < batch <= nv.MAX_SAMPLE_BATCH:
        raise ValueError(f"batch_size must be in [1, {nv.MAX_SAMPLE_BATCH}], got {batch}")
    check(nv.lib().cstr_replay_sample_mt19937_f32(C.byref(ring.c), ptr(ring.ctl), ptr(mt_state), C.c_int64(batch),
                                                  ptr(out_obs), ptr(out_act), ptr(out_next_obs), ptr(out_done),
                                                  ptr(out_rew), ptr(out_row_idx), ptr(out_env_idx), stream_ptr()),
          "cstr_replay_sample_mt19937_f32")


def vecnorm_init(vn_state):
    _chk(vn_state, "vn_state", (nv.VECNORM_STATE_WORDS,), th.float64)
    check(nv.lib().cstr_vecnorm_init_f64(ptr(vn_state), stream_ptr()), "cstr_vecnorm_init_f64")


def vecnorm_step(cfg, vn_state, returns, obs, reward, done, norm_obs_out=None, norm_reward_out=None):
    """VecNormalize.step_wait / reset (reward=None) on raw device tensors (reference: vec_normalize.py:174-204, :291-307)."""
    n, d = obs.shape
    if d != cfg.obs_dim:
        raise ValueError(f"obs has {d} columns, VecNormalize was built for {cfg.obs_dim}")
    _chk(vn_state, "vn_state", (nv.VECNORM_STATE_WORDS,), th.float64), _chk(returns, "returns", (n,), th.float64)
    _chk(obs, "obs", (n, d), th.float32)
    _opt(reward, "reward", (n,), th.float32), _opt(done, "done", (n,), th.float32)
    _opt(norm_obs_out, "norm_obs_out", (n, d), th.float32), _opt(norm_reward_out, "norm_reward_out", (n,), th.float32)
    if norm_obs_out is not None and norm_obs_out.data_ptr() == obs.data_ptr():
        raise ValueError("norm_obs_out must not alias obs (the raw observation stays available as get_original_obs)")
    if reward is None and (done is not None or norm_reward_out is not None):
        raise ValueError("the reset form (reward=None) takes neither done nor norm_reward_out")
    check(nv.lib().cstr_vecnorm_step_f64(C.byref(cfg), ptr(vn_state), ptr(returns), ptr(obs), ptr(reward), ptr(done),
                                         ptr(norm_obs_out), ptr(norm_reward_out), C.c_int64(n), stream_ptr()), "cstr_vecnorm_step_f64")


def vecnorm_apply(cfg, vn_state, obs, next_obs, reward):
    """normalize_obs / normalize_reward of a sampled batch, in place (reference: buffers.py:143-155, :312-323)."""
    _chk(vn_state, "vn_state", (nv.VECNORM_STATE_WORDS,), th.float64)
    ref = next((t for t in (obs, next_obs, reward) if t is not None), None)
    if ref is None:
        raise ValueError("nothing to normalise")
    b = ref.shape[0]
    _opt(obs, "obs", (b, cfg.obs_dim), th.float32), _opt(next_obs, "next_obs", (b, cfg.obs_dim), th.float32)
    if reward is not None and reward.numel() != b:
        raise ValueError(f"reward has {reward.numel()} entries, batch is {b}")
    _opt(reward, "reward", tuple(reward.shape) if reward is not None else (), th.float32)
    check(nv.lib().cstr_vecnorm_apply_f32(C.byref(cfg), ptr(vn_state), ptr(obs), ptr(next_obs), ptr(reward), C.c_int64(b),
                                          stream_ptr()), "cstr_vecnorm_apply_f32")


def replay_sample_packed(ring: DeviceRing, mt_state, batch: int, x_data, x_next, x_pi, out_done, out_rew, out_row_idx=None,
                         out_env_idx=None):
    """ReplayBuffer.sample gathered straight into the critic-input rows (obs | act), (next_obs | .), (obs | .)."""
    w = ring.obs_dim + ring.act_dim
    _chk(mt_state, "mt_state", (nv.MT_STATE_WORDS,), th.int32)
    _chk(x_data, "x_data", (batch, w), th.float32), _chk(x_next, "x_next", (batch, w), th.float32), _opt(x_pi, "x_pi", (batch, w), th.float32)
    _chk(out_done, "out_done", (batch, 1), th.float32), _chk(out_rew, "out_rew", (batch, 1), th.float32)
    _opt(out_row_idx, "out_row_idx", (batch,), th.int64), _opt(out_env_idx, "out_env_idx", (batch,), th.int64)
    if not 0 < batch <= nv.MAX_SAMPLE_BATCH:
        raise ValueError(f"batch_size must be in [1, {nv.MAX_SAMPLE_BATCH}], got {batch}")
    check(nv.lib().cstr_replay_sample_packed_mt19937_f32(C.byref(ring.c), ptr(ring.ctl), ptr(mt_state), C.c_int64(batch), ptr(x_data),
                                                         ptr(x_next), ptr(x_pi), ptr(out_done), ptr(out_rew), ptr(out_row_idx),
                                                         ptr(out_env_idx), stream_ptr()), "cstr_replay_sample_packed_mt19937_f32")


def replay_gather_packed(ring: DeviceRing, sample_idx, batch: int, x_data, x_next, x_pi, out_done, out_rew, out_row_idx=None,
                         out_env_idx=None, advance_ring: bool = False, rng_advance=None):
    """The gather of `replay_sample_packed` for index pairs drawn by `rollout_step` (sample_idx int32 [2, batch]), plus the control-word
    updates that launch left to this one: ReplayBuffer.add's epilogue (`advance_ring`) and `rng_advance` = (rng_ctl, count)."""
    w = ring.obs_dim + ring.act_dim
    _chk(sample_idx, "sample_idx", (2, batch), th.int32)
    _chk(x_data, "x_data", (batch, w), th.float32), _chk(x_next, "x_next", (batch, w), th.float32), _opt(x_pi, "x_pi", (batch, w), th.float32)
    _chk(out_done, "out_done", (batch, 1), th.float32), _chk(out_rew, "out_rew", (batch, 1), th.float32)
    _opt(out_row_idx, "out_row_idx", (batch,), th.int64), _opt(out_env_idx, "out_env_idx", (batch,), th.int64)
    if not 0 < batch <= nv.MAX_SAMPLE_BATCH:
        raise ValueError(f"batch_size must be in [1, {nv.MAX_SAMPLE_BATCH}], got {batch}")
    rng_ctl, rng_count = rng_advance if rng_advance is not None else (None, 0)
    _opt(rng_ctl, "rng_ctl", (nv.RNG_CTL_WORDS,), th.int64)
    check(nv.lib().cstr_replay_gather_packed_f32(C.byref(ring.c), ptr(ring.ctl), C.c_int(1 if advance_ring else 0), ptr(rng_ctl),
                                                 C.c_uint64(int(rng_count)), ptr(sample_idx), C.c_int64(batch), ptr(x_data), ptr(x_next),
                                                 ptr(x_pi), ptr(out_done), ptr(out_rew), ptr(out_row_idx), ptr(out_env_idx), stream_ptr()),
          "cstr_replay_gather_packed_f32")


def linear_act_fwd_gather(ring: DeviceRing, sample_idx, batch: int, both: bool, weight, bias, act: int, x_data, x_next, x_pi, out_done,
                          out_rew, advance_ring: bool = False, rng_advance=None, out=None):
    """`replay_gather_packed` fused into the first Linear layer behind it (cstr_linear_act_fwd_gather_f32): y = act(x W^T + b) with
    x = the sampled observations (rows [0, B)) and next observations (rows [B, 2B)) for `both`, else the B next observations; the
    packed batch is written for the later launches and the control words are advanced like `replay_gather_packed` does."""
    d, w = ring.obs_dim, ring.obs_dim + ring.act_dim
    n = weight.shape[0]
    m = 2 * batch if both else batch
    _chk(sample_idx, "sample_idx", (2, batch), th.int32), _chk(weight, "weight", (n, d), th.float32), _chk(bias, "bias", (n,), th.float32)
    _chk(x_data, "x_data", (batch, w), th.float32), _chk(x_next, "x_next", (batch, w), th.float32), _opt(x_pi, "x_pi", (batch, w), th.float32)
    _chk(out_done, "out_done", (batch, 1), th.float32), _chk(out_rew, "out_rew", (batch, 1), th.float32)
    if not 0 < batch <= nv.MAX_SAMPLE_BATCH:
        raise ValueError(f"batch_size must be in [1, {nv.MAX_SAMPLE_BATCH}], got {batch}")
    if out is None:
        out = th.empty(m, n, dtype=th.float32, device=weight.device)
    _chk(out, "out", (m, n), th.float32)
    rng_ctl, rng_count = rng_advance if rng_advance is not None else (None, 0)
    _opt(rng_ctl, "rng_ctl", (nv.RNG_CTL_WORDS,), th.int64)
    check(nv.lib().cstr_linear_act_fwd_gather_f32(C.byref(ring.c), ptr(ring.ctl), C.c_int(1 if advance_ring else 0), ptr(rng_ctl),
                                                  C.c_uint64(int(rng_count)), ptr(sample_idx), C.c_int64(batch), C.c_int(1 if both else 0),
                                                  ptr(weight), ptr(bias), C.c_int(act), ptr(out), C.c_int64(n), ptr(x_data), ptr(x_next),
                                                  ptr(x_pi), ptr(out_done), ptr(out_rew), stream_ptr()), "cstr_linear_act_fwd_gather_f32")
    return out


def td_target_min(q1, q2, logp, rew, done, ent_coef, gamma: float, out):
    n = q1.numel()
    for t, nm in ((q1, "q1"), (q2, "q2"), (rew, "rew"), (done, "done"), (out, "out")):
        if t.numel() != n:
            raise ValueError(f"{nm}: numel {t.numel()} != {n}")
        _chk(t, nm, t.shape, th.float32)
    if (logp is None) != (ent_coef is None):
        raise ValueError("logp and ent_coef go together (SAC) or are both None (TD3)")
    if logp is not None:
        _chk(logp, "logp", logp.shape, th.float32), _chk(ent_coef, "ent_coef", ent_coef.shape, th.float32)
        if logp.numel() != n or ent_coef.numel() != 1:
            raise ValueError("logp must have n elements and ent_coef exactly one")
    check(nv.lib().cstr_td_target_min_f32(ptr(q1), ptr(q2), ptr(logp), ptr(rew), ptr(done), ptr(ent_coef),
                                          C.c_float(gamma), ptr(out), C.c_int64(n), stream_ptr()), "cstr_td_target_min_f32")


def polyak(param_flat, target_flat, tau: float):
    n = param_flat.numel()
    _chk(param_flat, "param_flat", (n,), th.float32), _chk(target_flat, "target_flat", (n,), th.float32)
    check(nv.lib().cstr_polyak_f32(ptr(param_flat), ptr(target_flat), C.c_double(tau), C.c_int64(n), stream_ptr()),
          "cstr_polyak_f32")


def new_adam_ctl(device, step: int = 0, beta1: float = 0.9, beta2: float = 0.999) -> th.Tensor:
    """adam_ctl = {step, ticket, beta1^step, beta2^step} (the two powers stored as f64 bit patterns in the int64 tensor)."""
    ctl = th.zeros(nv.ADAM_CTL_WORDS, dtype=th.int64, device=device)
    set_adam_step(ctl, step, beta1, beta2)
    return ctl


def set_adam_step(ctl: th.Tensor, step: int, beta1: float = 0.9, beta2: float = 0.999) -> None:
    host = th.zeros(nv.ADAM_CTL_WORDS, dtype=th.int64)
    host[0] = int(step)
    host.view(th.float64)[2] = float(beta1) ** int(step)
    host.view(th.float64)[3] = float(beta2) ** int(step)
    ctl.copy_(host)


def adam(param, grad, exp_avg, exp_avg_sq, adam_ctl, lr_dev, beta1=0.9, beta2=0.999, eps=1e-8, grad_scale=1.0):
    n = param.numel()
    for t, nm in ((param, "param"), (grad, "grad"), (exp_avg, "exp_avg"), (exp_avg_sq, "exp_avg_sq")):
        _chk(t, nm, (n,), th.float32)
    _chk(adam_ctl, "adam_ctl", (nv.ADAM_CTL_WORDS,), th.int64), _chk(lr_dev, "lr", (1,), th.float64)
    check(nv.lib().cstr_adam_f32(ptr(param), ptr(grad), ptr(exp_avg), ptr(exp_avg_sq), ptr(adam_ctl), ptr(lr_dev),
                                 C.c_double(beta1), C.c_double(beta2), C.c_double(eps), C.c_float(grad_scale),
                                 C.c_int64(n), stream_ptr()), "cstr_adam_f32")


def _adam_segments(segments):
    """ctypes array of cstr_adam_seg_t from adam_multi's segment tuples."""
    segs = list(segments)
    arr = (nv.AdamSeg * max(len(segs), 1))()
    _fill_adam_segments(arr, segs)
    return arr, len(segs)


def adam_multi(segments):
    """Several flat-arena updates in one launch; `segments` = iterable of adam()'s positional argument tuples (optionally
    followed by a shadow = (tile-major copy tensor, begin, n, k) or None: see policy_swizzle, and by own_target = (target tensor,
    tau): the soft update of THESE parameters' target in the same pass), or ("polyak", source, target, tau) for a soft target
    update of parameters no segment changes (<= 4 segments, mutually independent)."""
    segs = list(segments)
    arr = (nv.AdamSeg * len(segs))()
    _fill_adam_segments(arr, segs)
    check(nv.lib().cstr_adam_multi_f32(arr, C.c_int(len(segs)), stream_ptr()), "cstr_adam_multi_f32")


def _fill_adam_segments(arr, segs) -> None:
    for i, seg in enumerate(segs):
        if seg[0] == "polyak":
            _, source, target, tau = seg
            n = source.numel()
            _chk(source, "source", (n,), th.float32), _chk(target, "target", (n,), th.float32)
            arr[i] = nv.AdamSeg(target.data_ptr(), None, None, None, None, None, 0.0, 0.0, 0.0, 1.0, n, source.data_ptr(), float(tau),
                                None, 0, 0, 0, None)
            continue
        param, grad, exp_avg, exp_avg_sq, adam_ctl, lr_dev, beta1, beta2, eps, grad_scale = seg[:10]
        shadow = seg[10] if len(seg) > 10 else None
        n = param.numel()
        for t, nm in ((param, "param"), (grad, "grad"), (exp_avg, "exp_avg"), (exp_avg_sq, "exp_avg_sq")):
            _chk(t, nm, (n,), th.float32)
        _chk(adam_ctl, "adam_ctl", (nv.ADAM_CTL_WORDS,), th.int64), _chk(lr_dev, "lr", (1,), th.float64)
        sh = (None, 0, 0, 0)
        if shadow is not None:
            t, begin, rows, cols = shadow
            if _f32c(t, "shadow").numel() != swizzled_numel(rows, cols) or begin % 4 or cols % 4 or begin + rows * cols > n:
                raise ValueError("shadow: wrong size or position")
            sh = (t.data_ptr(), begin, rows, cols)
        own, tau = (None, 0.0)
        if len(seg) > 11 and seg[11] is not None:
            own, tau = seg[11]
            _chk(own, "own_target", (n,), th.float32)
        arr[i] = nv.AdamSeg(param.data_ptr(), grad.data_ptr(), exp_avg.data_ptr(), exp_avg_sq.data_ptr(), adam_ctl.data_ptr(),
                            lr_dev.data_ptr(), beta1, beta2, eps, grad_scale, n, None, float(tau), *sh, None if own is None else own.data_ptr())


# ---- learner glue around the GEMMs (csrc/cstr_mlp.hip) ------------------------------------------------------------
ACT = {"none": 0, "relu": 1, "tanh": 2}


def _f32c(t, name):
    if not (isinstance(t, th.Tensor) and t.is_cuda and t.dtype == th.float32 and t.is_contiguous()):
        raise ValueError(f"{name}: needs a contiguous float32 device tensor")
    return t


def _gmn(t):
    """[m, n] -> (1, m, n); [G, m, n] -> (G, m, n)"""
    if t.dim() == 2:
        return (1, t.shape[0], t.shape[1])
    if t.dim() == 3:
        return tuple(t.shape)
    raise ValueError(f"expected a 2-D or 3-D tensor, got shape {tuple(t.shape)}")


def bias_act_fwd_(y, bias, act: int):
    g, m, n = _gmn(y)
    _f32c(y, "y"), _f32c(bias, "bias")
    if bias.numel() != g * n:
        raise ValueError(f"bias has {bias.numel()} elements, expected {g * n}")
    check(nv.lib().cstr_bias_act_fwd_f32(ptr(y), ptr(bias), C.c_int(act), C.c_int64(g), C.c_int64(m), C.c_int64(n), stream_ptr()),
          "cstr_bias_act_fwd_f32")
    return y


def bias_act_bwd(gy, y, act: int, gz, gbias):
    g, m, n = _gmn(gy)
    _f32c(gy, "gy"), _chk(gz, "gz", gy.shape, th.float32)
    if act != 0:
        _chk(y, "y", gy.shape, th.float32)
    if gbias is not None:
        _f32c(gbias, "gbias")
        if gbias.numel() != g * n:
            raise ValueError(f"gbias has {gbias.numel()} elements, expected {g * n}")
    check(nv.lib().cstr_bias_act_bwd_f32(ptr(gy), ptr(y), C.c_int(act), ptr(gz), ptr(gbias), C.c_int64(g), C.c_int64(m), C.c_int64(n),
                                         stream_ptr()), "cstr_bias_act_bwd_f32")


def bias_act_bwd_rows(gy, y, act: int, gz, gbias=None):
    """`bias_act_bwd` for one group whose gy / y may be column blocks of wider row-major matrices (row-strided views); gz is a
    contiguous [M, N] tensor."""
    m, n = gz.shape
    _chk(gz, "gz", (m, n), th.float32)
    ldg = _rows(gy, "gy", m, n)
    ldy = _rows(y, "y", m, n) if act != 0 else n
    if gbias is not None and _f32c(gbias, "gbias").numel() != n:
        raise ValueError(f"gbias has {gbias.numel()} elements, expected {n}")
    check(nv.lib().cstr_bias_act_bwd_rows_f32(ptr(gy), C.c_int64(ldg), ptr(y if act != 0 else None), C.c_int64(ldy), C.c_int(act), ptr(gz),
                                              ptr(gbias), C.c_int64(m), C.c_int64(n), stream_ptr()), "cstr_bias_act_bwd_rows_f32")
    return gz


def new_rng_ctl(seed: int, device) -> th.Tensor:
    """{seed, offset, ticket, -, sub-tickets[8], -} of the in-kernel Philox stream (cstr_gaussian_head_fwd_f32), int64 bit patterns"""
    return th.tensor([int(seed) & 0x7FFFFFFFFFFFFFFF] + [0] * (nv.RNG_CTL_WORDS - 1), dtype=th.int64).to(device)


def gaussian_head_fwd_(params, bias, eps, rng_ctl, action, logp):
    """params [B, 2A] += bias in place; action [B, A] (row-strided view allowed) = tanh(mean + std * eps); eps is drawn in
    the kernel when rng_ctl is given (and stored for the backward), read otherwise."""
    b, a2 = params.shape
    a = a2 // 2
    _chk(params, "params", (b, 2 * a), th.float32), _chk(eps, "eps", (b, a), th.float32)
    if bias is not None:
        _chk(bias, "bias", (2 * a,), th.float32)
    _opt(rng_ctl, "rng_ctl", (nv.RNG_CTL_WORDS,), th.int64), _opt(logp, "logp", (b,), th.float32)
    stride = _rows(action, "action", b, a)
    if a > nv.MAX_HEAD_ACT:
        raise ValueError(f"gaussian head supports up to {nv.MAX_HEAD_ACT} action dimensions, got {a}")
    check(nv.lib().cstr_gaussian_head_fwd_f32(ptr(params), ptr(bias), ptr(eps), ptr(rng_ctl), ptr(action), C.c_int64(stride), ptr(logp),
                                              C.c_int64(b), C.c_int(a), stream_ptr()), "cstr_gaussian_head_fwd_f32")


def gaussian_head_gemm_fwd(hidden, weight, bias, params, eps, rng_ctl, action, logp):
    """gaussian_head_fwd_ with the head's Linear inside: params [B, 2A] = hidden @ weight^T + bias is an OUTPUT."""
    b, k = hidden.shape
    a2 = weight.shape[0]
    a = a2 // 2
    if not (hidden.is_cuda and hidden.dtype == th.float32 and hidden.stride(1) == 1):
        raise ValueError("hidden: needs a float32 device matrix with unit column stride")
    _chk(weight, "weight", (a2, k), th.float32), _chk(bias, "bias", (a2,), th.float32)
    _chk(params, "params", (b, a2), th.float32), _chk(eps, "eps", (b, a), th.float32)
    _opt(rng_ctl, "rng_ctl", (nv.RNG_CTL_WORDS,), th.int64), _opt(logp, "logp", (b,), th.float32)
    stride = _rows(action, "action", b, a)
    check(nv.lib().cstr_gaussian_head_gemm_fwd_f32(ptr(hidden), C.c_int64(hidden.stride(0)), ptr(weight), ptr(bias), ptr(params), ptr(eps),
                                                   ptr(rng_ctl), ptr(action), C.c_int64(stride), ptr(logp), C.c_int64(b), C.c_int(a),
                                                   C.c_int64(k), stream_ptr()), "cstr_gaussian_head_gemm_fwd_f32")


def gaussian_head_bwd(g_action, g_logp, action, params, eps, g_params, g_bias):
    b, a2 = params.shape
    a = a2 // 2
    _chk(params, "params", (b, 2 * a), th.float32), _chk(eps, "eps", (b, a), th.float32), _chk(g_params, "g_params", (b, 2 * a), th.float32)
    _opt(g_logp, "g_logp", (b,), th.float32), _opt(g_bias, "g_bias", (2 * a,), th.float32)
    stride = _rows(action, "action", b, a)
    ga_stride = 0 if g_action is None else _rows(g_action, "g_action", b, a)
    check(nv.lib().cstr_gaussian_head_bwd_f32(ptr(g_action), C.c_int64(ga_stride), ptr(g_logp), ptr(action), C.c_int64(stride),
                                              ptr(params), ptr(eps), ptr(g_params), ptr(g_bias), C.c_int64(b), C.c_int(a), stream_ptr()),
          "cstr_gaussian_head_bwd_f32")


def gaussian_head_bwd_input(g_action, g_logp, action, params, eps, weight, hidden, act: int, g_params, dz):
    """g_params (as gaussian_head_bwd) and dz = (g_params @ weight) * act'(hidden) in one launch; weight [2A, H], hidden [B, H]."""
    b, a2 = params.shape
    a = a2 // 2
    h = weight.shape[1]
    _chk(params, "params", (b, 2 * a), th.float32), _chk(eps, "eps", (b, a), th.float32), _chk(g_params, "g_params", (b, 2 * a), th.float32)
    _chk(weight, "weight", (2 * a, h), th.float32), _chk(dz, "dz", (b, h), th.float32), _opt(g_logp, "g_logp", (b,), th.float32)
    stride = _rows(action, "action", b, a)
    ga_stride = 0 if g_action is None else _rows(g_action, "g_action", b, a)
    ldh = _rows(hidden, "hidden", b, h)
    check(nv.lib().cstr_gaussian_head_bwd_input_f32(ptr(g_action), C.c_int64(ga_stride), ptr(g_logp), ptr(action), C.c_int64(stride),
                                                    ptr(params), ptr(eps), ptr(weight), ptr(hidden), C.c_int64(ldh), C.c_int(act),
                                                    ptr(g_params), ptr(dz), C.c_int64(b), C.c_int(a), C.c_int64(h), stream_ptr()),
          "cstr_gaussian_head_bwd_input_f32")
    return dz


POLICY_LDS_FLOATS = 16 * 1024  # 64 KB


def swizzled_numel(n: int, k: int) -> int:
    return -(-n // 16) * -(-k // 16) * 256


def policy_swizzle(w, out=None):
    """Tile-major copy of a weight matrix [N, K] (K % 4 == 0) for the policy kernel's matrix-core operand loads
    (cstr_policy_swizzle_f32): a wave's load then reads 1 KB of consecutive bytes."""
    n, k = w.shape
    _chk(w, "w", (n, k), th.float32)
    if out is None:
        out = th.empty(swizzled_numel(n, k), dtype=th.float32, device=w.device)
    _chk(out, "out", (swizzled_numel(n, k),), th.float32)
    check(nv.lib().cstr_policy_swizzle_f32(ptr(w), C.c_int64(n), C.c_int64(k), ptr(out), stream_ptr()), "cstr_policy_swizzle_f32")
    return out


def policy_rows_supported(k0: int, h1: int, h2: int, n_out: int) -> bool:
    return (k0 <= 256 and h1 % 4 == 0 and h2 % 4 == 0 and 16 * (h1 + h2 + 8) <= POLICY_LDS_FLOATS
            and n_out <= 2 * nv.MAX_HEAD_ACT)


def policy_rows_fwd(x, w1, b1, w2, b2, w3, b3, act: int, head: int, out_act: int, action, eps=None, rng_ctl=None, logp=None, w2_swz=None,
                    defer_rng_advance: bool = False):
    """A two-hidden-layer policy network + action head for all rows of x in ONE launch, inference only (cstr_policy_rows_fwd_f32).
    head 0: squashed-Gaussian sample (w3 [2A, H2]; noise from `eps` [M, A] or the Philox stream `rng_ctl`; optional `logp`);
    head 1: deterministic out_act(h2 @ w3^T + b3). `action` [M, A] may be a column block of a wider row-major matrix.
    `defer_rng_advance`: the launch leaves the stream offset alone -- the caller advances it by M before the next consumer
    (collect_step(..., rng_advance=(rng_ctl, M)) does it in the collect kernel's last-workgroup epilogue)."""
    m, k0 = x.shape
    h1, h2, n_out = w1.shape[0], w2.shape[0], w3.shape[0]
    a = n_out // 2 if head == 0 else n_out
    if not (x.is_cuda and x.dtype == th.float32 and x.dim() == 2 and x.stride(1) == 1):
        raise ValueError("x: needs a float32 device matrix with unit column stride")
    _chk(w1, "w1", (h1, k0), th.float32), _chk(b1, "b1", (h1,), th.float32), _chk(w2, "w2", (h2, h1), th.float32)
    _chk(b2, "b2", (h2,), th.float32), _chk(w3, "w3", (n_out, h2), th.float32), _chk(b3, "b3", (n_out,), th.float32)
    _opt(eps, "eps", (m, a), th.float32), _opt(rng_ctl, "rng_ctl", (nv.RNG_CTL_WORDS,), th.int64), _opt(logp, "logp", (m,), th.float32)
    stride = _rows(action, "action", m, a)
    if w2_swz is not None:
        _chk(w2_swz, "w2_swz", (swizzled_numel(h2, h1),), th.float32)
    net = nv.PolicyMlp(k0, h1, h2, a, act, head, out_act, 1 if (defer_rng_advance and rng_ctl is not None) else 0, w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr(), w3.data_ptr(),
                       b3.data_ptr(), None if w2_swz is None else w2_swz.data_ptr())
    check(nv.lib().cstr_policy_rows_fwd_f32(C.byref(net), ptr(x), C.c_int64(max(x.stride(0), k0)), ptr(eps), ptr(rng_ctl), ptr(action),
                                            C.c_int64(stride), ptr(logp), C.c_int64(m), stream_ptr()), "cstr_policy_rows_fwd_f32")
    return action


def rollout_step_supported(k0: int, h1: int, h2: int, n_out: int, has_swizzled_w2: bool) -> bool:
    """cstr_rollout_step_f32 covers the pipelined policy kernel with a one-chunk first layer (see the header)."""
    return has_swizzled_w2 and k0 <= 16 and k0 % 4 == 0 and h1 <= 512 and h2 <= 512 and policy_rows_supported(k0, h1, h2, n_out)


def rollout_step(x, w1, b1, w2, b2, w3, b3, act: int, head: int, out_act: int, w2_swz, rng_ctl, coef, integrator: str, ring: DeviceRing,
                 env_obs, step_count, squashed, act_low, act_high, noise=None, reset_obs=None, pcg_state=None, static_init=None,
                 reward_out=None, done_out=None, ep_return=None, ep_stats=None, action_out=None, mt_state=None, sample_idx=None):
    """One vec-step in ONE launch (cstr_rollout_step_f32): policy network + sampling on x [N, k0], the fused collect step of
    `collect_step` with the action kept in registers, and -- when `mt_state` / `sample_idx` int32 [2, batch] are given -- the replay
    index draw of the gradient step behind it. Writes NO control word: follow it with `replay_gather_packed(advance_ring=True,
    rng_advance=(rng_ctl, N))`."""
    n, d, a = ring.n_envs, ring.obs_dim, ring.act_dim
    k0 = x.shape[1]
    h1, h2, n_out = w1.shape[0], w2.shape[0], w3.shape[0]
    if not (x.is_cuda and x.dtype == th.float32 and x.dim() == 2 and x.shape[0] == n and x.stride(1) == 1):
        raise ValueError("x: needs a float32 device matrix [n_envs, k0] with unit column stride")
    if (n_out // 2 if head == 0 else n_out) != a:
        raise ValueError(f"policy head width {n_out} does not match the ring's act_dim {a}")
    _chk(w1, "w1", (h1, k0), th.float32), _chk(b1, "b1", (h1,), th.float32), _chk(w2, "w2", (h2, h1), th.float32)
    _chk(b2, "b2", (h2,), th.float32), _chk(w3, "w3", (n_out, h2), th.float32), _chk(b3, "b3", (n_out,), th.float32)
    _chk(w2_swz, "w2_swz", (swizzled_numel(h2, h1),), th.float32)
    _opt(rng_ctl, "rng_ctl", (nv.RNG_CTL_WORDS,), th.int64)
    _chk(env_obs, "env_obs", (n, d), th.float32), _chk(step_count, "step_count", (n,), th.int32)
    _opt(noise, "noise", (n, a), th.float32), _opt(reset_obs, "reset_obs", (n, d), th.float32)
    _opt(pcg_state, "pcg_state", (n, nv.PCG_STATE_WORDS), th.int64)
    _opt(static_init, "static_init", (n, 8 if a == 4 else 4), th.float64)
    _opt(reward_out, "reward_out", (n,), th.float32), _opt(done_out, "done_out", (n,), th.float32)
    _opt(ep_return, "ep_return", (n,), th.float32), _opt(ep_stats, "ep_stats", (4,), th.float64)
    _opt(action_out, "action_out", (n, a), th.float32)
    if (ep_return is None) != (ep_stats is None):
        raise ValueError("ep_return and ep_stats go together")
    if (reset_obs is None) == (pcg_state is None):
        raise ValueError("rollout_step needs exactly one reset source: reset_obs or pcg_state")
    if len(act_low) != a or len(act_high) != a:
        raise ValueError(f"act_low/act_high need {a} entries")
    if (mt_state is None) != (sample_idx is None):
        raise ValueError("mt_state and sample_idx go together")
    batch = 0
    if mt_state is not None:
        _chk(mt_state, "mt_state", (nv.MT_STATE_WORDS,), th.int32)
        batch = sample_idx.shape[1] if sample_idx.dim() == 2 else -1
        _chk(sample_idx, "sample_idx", (2, batch), th.int32)
    lo = (C.c_float * a)(*[float(v) for v in act_low])
    hi = (C.c_float * a)(*[float(v) for v in act_high])
    net = nv.PolicyMlp(k0, h1, h2, a, act, head, out_act, 1, w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr(), w3.data_ptr(),
                       b3.data_ptr(), w2_swz.data_ptr())
    check(nv.lib().cstr_rollout_step_f32(C.byref(net), ptr(x), C.c_int64(max(x.stride(0), k0)), ptr(rng_ctl), C.byref(coef),
                                         C.c_int(INTEGRATORS[integrator]), C.byref(ring.c), ptr(ring.ctl), ptr(env_obs), ptr(step_count),
                                         C.c_int(int(squashed)), lo, hi, ptr(noise), ptr(reset_obs), ptr(pcg_state), ptr(static_init),
                                         ptr(reward_out), ptr(done_out), ptr(ep_return), ptr(ep_stats), ptr(action_out), ptr(mt_state),
                                         C.c_int64(batch), ptr(sample_idx), stream_ptr()), "cstr_rollout_step_f32")


def linear_act_fwd(x, weight, bias, act: int, out=None):
    """y = act(x @ W^T + b) in one launch (f32 matrix cores). x [M, K] or [G, M, K] with unit inner stride (rows / groups may
    be strided, a stride-0 group dimension shares the input), weight [N, K] / [G, N, K], bias [N] / [G, N] contiguous."""
    if x.dim() == 2:
        g, (m, k), gs, ldx = 1, x.shape, 0, x.stride(0)
        n = weight.shape[0]
        shape = (m, n)
    else:
        g, m, k = x.shape
        gs, ldx = x.stride(0), x.stride(1)
        n = weight.shape[1]
        shape = (g, m, n)
    if not (x.is_cuda and x.dtype == th.float32 and x.stride(-1) == 1 and ldx >= k):
        raise ValueError("x: needs a float32 device tensor with unit inner stride")
    if m == 1:
        ldx = max(ldx, k)
    _f32c(weight, "weight"), _f32c(bias, "bias")
    if weight.numel() != g * n * k or bias.numel() != g * n:
        raise ValueError(f"weight / bias do not match x {tuple(x.shape)}: {tuple(weight.shape)}, {tuple(bias.shape)}")
    y = th.empty(shape, dtype=th.float32, device=x.device) if out is None else _chk(out, "out", shape, th.float32)
    check(nv.lib().cstr_linear_act_fwd_f32(ptr(x), C.c_int64(gs), C.c_int64(ldx), ptr(weight), ptr(bias), C.c_int(act), ptr(y),
                                           C.c_int64(g), C.c_int64(m), C.c_int64(n), C.c_int64(k), stream_ptr()), "cstr_linear_act_fwd_f32")
    return y


def linear_act_fwd_sets(sets, act: int):
    """Several independent Linear + bias + activation layers of ONE shape in one launch. `sets`: [(x [M, K] (row-strided ok),
    weight [N, K], bias [N], y [M, N] (row-strided ok: e.g. a column block of the joint action))]."""
    sets = list(sets)
    if not 0 < len(sets) <= nv.MAX_LINEAR_SETS:
        raise ValueError(f"1..{nv.MAX_LINEAR_SETS} sets, got {len(sets)}")
    m, k = sets[0][0].shape
    n = sets[0][1].shape[0]
    arr = (nv.LinearSet * len(sets))()
    for i, (x, w, b, y) in enumerate(sets):
        for t, nm, shape in ((x, "x", (m, k)), (y, "y", (m, n))):
            if not (t.is_cuda and t.dtype == th.float32 and tuple(t.shape) == shape and t.stride(1) == 1):
                raise ValueError(f"{nm}[{i}]: needs a float32 device matrix {shape} with unit column stride")
        _chk(w, f"weight[{i}]", (n, k), th.float32), _chk(b, f"bias[{i}]", (n,), th.float32)
        arr[i] = nv.LinearSet(x.data_ptr(), max(x.stride(0), k), w.data_ptr(), b.data_ptr(), y.data_ptr(), max(y.stride(0), n))
    check(nv.lib().cstr_linear_act_fwd_sets_f32(arr, C.c_int(len(sets)), C.c_int(act), C.c_int64(m), C.c_int64(n), C.c_int64(k),
                                                stream_ptr()), "cstr_linear_act_fwd_sets_f32")


def linear_bwd_input(gz, weight, y, act: int, sum_groups: bool = False):
    """dz = (gz @ W) * act'(y): a Linear's input gradient fused with the activation gradient of the layer below (whose output
    y is this layer's input). gz [M, N] / [G, M, N], weight [N, K] / [G, N, K] contiguous. sum_groups: the G groups share one
    input, the result is the [M, K] sum over groups."""
    g, m, n = _gmn(gz)
    k = weight.shape[-1]
    _f32c(gz, "gz"), _f32c(weight, "weight")
    if weight.numel() != g * n * k:
        raise ValueError(f"weight {tuple(weight.shape)} does not match gz {tuple(gz.shape)}")
    shape = (m, k) if (gz.dim() == 2 or sum_groups) else (g, m, k)
    if act != 0:
        _chk(y, "y", shape, th.float32)
    dz = th.empty(shape, dtype=th.float32, device=gz.device)
    check(nv.lib().cstr_linear_bwd_input_f32(ptr(gz), ptr(weight), ptr(y if act != 0 else None), C.c_int(act), ptr(dz), C.c_int64(g),
                                             C.c_int(int(sum_groups)), C.c_int64(m), C.c_int64(n), C.c_int64(k), stream_ptr()),
          "cstr_linear_bwd_input_f32")
    return dz


def linear_bwd_weight(dz, x, dw, db=None):
    """dw = dz^T @ x and (optionally) db = column sums of dz in one launch. dz [M, N] / [G, M, N] contiguous; x [M, K] /
    [G, M, K] with unit inner stride (rows / groups may be strided, stride-0 groups share the input); dw, db: contiguous
    views of the gradient arena."""
    g, m, n = _gmn(dz)
    k = x.shape[-1]
    _f32c(dz, "dz"), _f32c(dw, "dw")
    if x.dim() == 2:
        gs, ldx = 0, x.stride(0)
    else:
        gs, ldx = x.stride(0), x.stride(1)
    if not (x.is_cuda and x.dtype == th.float32 and x.stride(-1) == 1 and x.shape[-2] == m):
        raise ValueError("x: needs a float32 device tensor with unit inner stride and as many rows as dz")
    if m == 1:
        ldx = max(ldx, k)
    if dw.numel() != g * n * k:
        raise ValueError(f"dw has {dw.numel()} elements, expected {g * n * k}")
    if db is not None and _f32c(db, "db").numel() != g * n:
        raise ValueError(f"db has {db.numel()} elements, expected {g * n}")
    check(nv.lib().cstr_linear_bwd_weight_f32(ptr(dz), ptr(x), C.c_int64(gs), C.c_int64(ldx), ptr(dw), ptr(db), C.c_int64(g),
                                              C.c_int64(m), C.c_int64(n), C.c_int64(k), stream_ptr()), "cstr_linear_bwd_weight_f32")


def linear_bwd_weight_sets(sets):
    """`linear_bwd_weight` for several independent (2-D) Linears of any shapes in one launch. `sets`: [(dz [M, N], x [M, K]
    (row-strided ok), dw [N, K] view, db [N] view or None)]."""
    sets = list(sets)
    if not 0 < len(sets) <= nv.MAX_LINEAR_SETS:
        raise ValueError(f"1..{nv.MAX_LINEAR_SETS} sets, got {len(sets)}")
    arr = (nv.WgradSet * len(sets))()
    for i, (dz, x, dw, db) in enumerate(sets):
        if dz.dim() != 2:
            raise ValueError(f"dz[{i}]: needs a matrix")
        m, n = dz.shape
        k = x.shape[-1]
        _f32c(dz, f"dz[{i}]"), _f32c(dw, f"dw[{i}]")
        if not (x.is_cuda and x.dtype == th.float32 and x.dim() == 2 and x.stride(1) == 1 and x.shape[0] == m):
            raise ValueError(f"x[{i}]: needs a float32 device matrix with unit inner stride and as many rows as dz")
        if dw.numel() != n * k or (db is not None and _f32c(db, f"db[{i}]").numel() != n):
            raise ValueError(f"dw / db[{i}] do not match dz {tuple(dz.shape)} and x {tuple(x.shape)}")
        arr[i] = nv.WgradSet(dz.data_ptr(), x.data_ptr(), max(x.stride(0), k), dw.data_ptr(), None if db is None else db.data_ptr(),
                             m, n, k)
    check(nv.lib().cstr_linear_bwd_weight_sets_f32(arr, C.c_int(len(sets)), stream_ptr()), "cstr_linear_bwd_weight_sets_f32")


def target_smooth(action, noise, rng_ctl, sigma: float, clip: float, out):
    """out = clamp(action + clamp(noise, -clip, clip), -1, 1); noise given ([B, A], already scaled) or drawn (rng_ctl)."""
    b, a = action.shape
    _chk(action, "action", (b, a), th.float32), _opt(noise, "noise", (b, a), th.float32)
    _opt(rng_ctl, "rng_ctl", (nv.RNG_CTL_WORDS,), th.int64)
    if (noise is None) == (rng_ctl is None):
        raise ValueError("target_smooth needs exactly one noise source: a noise tensor or rng_ctl")
    stride = _rows(out, "out", b, a)
    check(nv.lib().cstr_target_smooth_f32(ptr(action), ptr(noise), ptr(rng_ctl), C.c_float(sigma), C.c_float(clip), ptr(out),
                                          C.c_int64(stride), C.c_int64(b), C.c_int(a), stream_ptr()), "cstr_target_smooth_f32")
    return out


def linear_smooth_supported(m: int, n: int, k: int) -> bool:
    return n <= 16 and k > 32 and m <= 32768


def linear_smooth_fwd(x, weight, bias, act: int, noise, rng_ctl, sigma: float, clip: float, out):
    """`linear_act_fwd` followed by `target_smooth` in ONE launch (cstr_linear_smooth_fwd_f32): out [M, N] (may be a column block of a
    wider row-major matrix) = clamp(act(x W^T + b) + clamp(noise | sigma N(0, 1) from rng_ctl, -clip, clip), -1, 1)."""
    m, k = x.shape
    n = weight.shape[0]
    if not (x.is_cuda and x.dtype == th.float32 and x.dim() == 2 and x.stride(1) == 1):
        raise ValueError("x: needs a float32 device matrix with unit column stride")
    _chk(weight, "weight", (n, k), th.float32), _chk(bias, "bias", (n,), th.float32)
    _opt(noise, "noise", (m, n), th.float32), _opt(rng_ctl, "rng_ctl", (nv.RNG_CTL_WORDS,), th.int64)
    if (noise is None) == (rng_ctl is None):
        raise ValueError("exactly one noise source: noise or rng_ctl")
    stride = _rows(out, "out", m, n)
    check(nv.lib().cstr_linear_smooth_fwd_f32(ptr(x), C.c_int64(max(x.stride(0), k)), ptr(weight), ptr(bias), C.c_int(act), ptr(noise),
                                              ptr(rng_ctl), C.c_float(sigma), C.c_float(clip), ptr(out), C.c_int64(stride), C.c_int64(m),
                                              C.c_int64(n), C.c_int64(k), stream_ptr()), "cstr_linear_smooth_fwd_f32")
    return out


def hidden_head_fwd_(z, b1, act: int, w2, b2, q):
    """y = act(z + b1) in place on z [G, m, k] (or [m, k]); q [G, m, 1] = y . w2 + b2 (a Q network's scalar head)."""
    g, m, k = _gmn(z)
    for t, nm, numel in ((z, "z", g * m * k), (b1, "b1", g * k), (w2, "w2", g * k), (b2, "b2", g), (q, "q", g * m)):
        if _f32c(t, nm).numel() != numel:
            raise ValueError(f"{nm} has {t.numel()} elements, expected {numel}")
    check(nv.lib().cstr_hidden_head_fwd_f32(ptr(z), ptr(b1), C.c_int(act), ptr(w2), ptr(b2), ptr(q), C.c_int64(g), C.c_int64(m),
                                            C.c_int64(k), stream_ptr()), "cstr_hidden_head_fwd_f32")
    return q


def hidden_head_bwd(gq, y, act: int, w2, dz, gb1=None, gw2=None, gb2=None):
    """dz = gq * w2 * act'(y); with parameter gradients: gb1 = sum_m dz, gw2 = sum_m gq * y, gb2 = sum_m gq."""
    g, m, k = _gmn(y)
    _chk(dz, "dz", y.shape, th.float32)
    for t, nm, numel in ((gq, "gq", g * m), (y, "y", g * m * k), (w2, "w2", g * k)):
        if _f32c(t, nm).numel() != numel:
            raise ValueError(f"{nm} has {t.numel()} elements, expected {numel}")
    grads = (gb1, gw2, gb2)
    if any(t is None for t in grads) != all(t is None for t in grads):
        raise ValueError("gb1, gw2 and gb2 go together")
    if gb1 is not None:
        for t, nm, numel in ((gb1, "gb1", g * k), (gw2, "gw2", g * k), (gb2, "gb2", g)):
            if _f32c(t, nm).numel() != numel:
                raise ValueError(f"{nm} has {t.numel()} elements, expected {numel}")
    check(nv.lib().cstr_hidden_head_bwd_f32(ptr(gq), ptr(y), C.c_int(act), ptr(w2), ptr(dz), ptr(gb1), ptr(gw2), ptr(gb2),
                                            C.c_int64(g), C.c_int64(m), C.c_int64(k), stream_ptr()), "cstr_hidden_head_bwd_f32")


def hidden_head_bwd_root(root: dict, y, act: int, w2, dz, gb1=None, gw2=None, gb2=None):
    """`hidden_head_bwd` for the twin Q networks with the loss root inside (cstr_hidden_head_bwd_root_f32): `root` =
    dict(mode="td", q1_t, q2_t, next_logp | None, rew, done, ent_coef | None, gamma, scale, q1, q2, target_out | None, loss_out | None,
    loss_sum | None, alpha | None) -- the arguments of `td_twin_q_loss` -- or dict(mode="sac_actor", logp, q1, q2, ent_coef, g_logp,
    loss_out | None, loss_sum | None) -- those of `sac_actor_loss` --, or dict(mode="neg_mean", q1, loss_out | None, loss_sum | None): the
    deterministic actors' -mean(Q1) through the FIRST Q network alone (y [1, m, k])."""
    g, m, k = _gmn(y)
    if g != (1 if root["mode"] == "neg_mean" else 2):
        raise ValueError("the loss-root form is built for the twin Q networks (2 groups; mode 'neg_mean': the first one alone)")
    _chk(dz, "dz", y.shape, th.float32)
    for t, nm, numel in ((y, "y", g * m * k), (w2, "w2", g * k)):
        if _f32c(t, nm).numel() != numel:
            raise ValueError(f"{nm} has {t.numel()} elements, expected {numel}")
    grads = (gb1, gw2, gb2)
    if any(t is None for t in grads) != all(t is None for t in grads):
        raise ValueError("gb1, gw2 and gb2 go together")
    if gb1 is not None:
        for t, nm, numel in ((gb1, "gb1", g * k), (gw2, "gw2", g * k), (gb2, "gb2", g)):
            if _f32c(t, nm).numel() != numel:
                raise ValueError(f"{nm} has {t.numel()} elements, expected {numel}")
    p = lambda key, n=m: None if root.get(key) is None else _vec(root[key], key, n).data_ptr()  # noqa: E731
    part = nv.AlphaPart()
    if root["mode"] == "td":
        alpha = root.get("alpha")
        if alpha is not None:
            _vec(alpha["logp_pi"], "logp_pi", m)
            for nm in ("log_alpha", "grad_out", "ent_coef_out"):
                _vec(alpha[nm], nm, 1)
            part = nv.AlphaPart(alpha["log_alpha"].data_ptr(), alpha["logp_pi"].data_ptr(), float(alpha["target_entropy"]),
                                alpha["grad_out"].data_ptr(), alpha["ent_coef_out"].data_ptr(),
                                *(None if alpha.get(kk) is None else alpha[kk].data_ptr() for kk in ("loss_out", "loss_sum", "ent_coef_sum")))
        rt = nv.HeadRoot(1, m, float(root["gamma"]), float(root["scale"]), p("q1_t"), p("q2_t"), p("next_logp"), p("rew"), p("done"),
                         None if alpha is not None else p("ent_coef", 1), p("q1"), p("q2"), p("target_out"), None, None,
                         p("loss_out", 1), p("loss_sum", 1), part)
    elif root["mode"] == "sac_actor":
        rt = nv.HeadRoot(2, m, 0.0, 0.0, None, None, None, None, None, p("ent_coef", 1), p("q1"), p("q2"), None, p("logp"), p("g_logp"),
                         p("loss_out", 1), p("loss_sum", 1), part)
    elif root["mode"] == "neg_mean":  # neg_mean_loss's arguments: loss = -mean(q1)
        rt = nv.HeadRoot(3, m, 0.0, 0.0, None, None, None, None, None, None, p("q1"), None, None, None, None, p("loss_out", 1), p("loss_sum", 1),
                         part)
    else:
        raise ValueError(f"unknown loss root {root['mode']!r}")
    check(nv.lib().cstr_hidden_head_bwd_root_f32(C.byref(rt), ptr(y), C.c_int(act), ptr(w2), ptr(dz), ptr(gb1), ptr(gw2), ptr(gb2),
                                                 C.c_int64(m), C.c_int64(k), stream_ptr()), "cstr_hidden_head_bwd_root_f32")


def _rows(t, name, b, a):
    """A [b, a] float32 device matrix whose rows may be strided (a column slice of a wider row-major matrix)."""
    if not (isinstance(t, th.Tensor) and t.is_cuda and t.dtype == th.float32 and tuple(t.shape) == (b, a) and t.stride(1) == 1
            and t.stride(0) >= a):
        raise ValueError(f"{name}: needs a float32 device matrix [{b}, {a}] with unit column stride")
    return t.stride(0)


def squashed_gaussian_fwd(mean, log_std_raw, eps, action, logp):
    b, a = mean.shape
    stride = _rows(mean, "mean", b, a)
    if _rows(log_std_raw, "log_std_raw", b, a) != stride:
        raise ValueError("mean and log_std_raw must share their row stride")
    _chk(eps, "eps", (b, a), th.float32), _chk(action, "action", (b, a), th.float32)
    _opt(logp, "logp", (b,), th.float32)
    check(nv.lib().cstr_squashed_gaussian_fwd_f32(ptr(mean), ptr(log_std_raw), ptr(eps), ptr(action), ptr(logp), C.c_int64(b),
                                                  C.c_int(a), C.c_int(stride), stream_ptr()), "cstr_squashed_gaussian_fwd_f32")


def squashed_gaussian_bwd(g_action, g_logp, action, log_std_raw, eps, g_mean, g_log_std_raw):
    b, a = action.shape
    _chk(action, "action", (b, a), th.float32), _chk(eps, "eps", (b, a), th.float32)
    stride = _rows(log_std_raw, "log_std_raw", b, a)
    if _rows(g_mean, "g_mean", b, a) != stride or _rows(g_log_std_raw, "g_log_std_raw", b, a) != stride:
        raise ValueError("g_mean / g_log_std_raw must have log_std_raw's row stride")
    ga_stride = a if g_action is None else _rows(g_action, "g_action", b, a)
    if g_logp is not None and (g_logp.numel() != b or not g_logp.is_contiguous() or g_logp.dtype != th.float32):
        raise ValueError("g_logp must be a contiguous float32 tensor with batch elements")
    check(nv.lib().cstr_squashed_gaussian_bwd_f32(ptr(g_action), ptr(g_logp), ptr(action), ptr(log_std_raw), ptr(eps), ptr(g_mean),
                                                  ptr(g_log_std_raw), C.c_int64(b), C.c_int(a), C.c_int(stride), C.c_int(ga_stride),
                                                  stream_ptr()), "cstr_squashed_gaussian_bwd_f32")


def _vec(t, name, n):
    if not (isinstance(t, th.Tensor) and t.is_cuda and t.dtype == th.float32 and t.is_contiguous() and t.numel() == n):
        raise ValueError(f"{name}: needs a contiguous float32 device tensor with {n} elements")
    return t


def sac_alpha(log_alpha, logp, target_entropy: float, grad_out, ent_coef_out, loss_sum=None, ent_coef_sum=None, loss_out=None):
    b = logp.numel()
    _vec(logp, "logp", b)
    for t, nm in ((log_alpha, "log_alpha"), (grad_out, "grad_out"), (ent_coef_out, "ent_coef_out")):
        _vec(t, nm, 1)
    check(nv.lib().cstr_sac_alpha_f32(ptr(log_alpha), ptr(logp), C.c_float(target_entropy), ptr(grad_out), ptr(ent_coef_out),
                                      ptr(loss_out), ptr(loss_sum), ptr(ent_coef_sum), C.c_int64(b), stream_ptr()), "cstr_sac_alpha_f32")


def twin_q_loss(q1, q2, target, scale: float, gq1, gq2, loss_out=None, loss_sum=None):
    b = q1.numel()
    for t, nm in ((q1, "q1"), (q2, "q2"), (target, "target"), (gq1, "gq1"), (gq2, "gq2")):
        _vec(t, nm, b)
    check(nv.lib().cstr_twin_q_loss_f32(ptr(q1), ptr(q2), ptr(target), C.c_float(scale), ptr(gq1), ptr(gq2), ptr(loss_out),
                                        ptr(loss_sum), C.c_int64(b), stream_ptr()), "cstr_twin_q_loss_f32")


def td_twin_q_loss(q1_t, q2_t, next_logp, rew, done, ent_coef, gamma: float, q1, q2, scale: float, target_out, gq1, gq2,
                   loss_out=None, loss_sum=None, alpha=None):
    """`td_target_min` + `twin_q_loss` (+ `sac_alpha`) in one launch. `alpha` = None or a dict(log_alpha, logp_pi,
    target_entropy, grad_out, ent_coef_out, loss_out=None, loss_sum=None, ent_coef_sum=None): SAC's entropy-coefficient loss
    rides along and the target uses exp(log_alpha) (`ent_coef` is ignored then)."""
    b = q1.numel()
    for t, nm in ((q1_t, "q1_t"), (q2_t, "q2_t"), (rew, "rew"), (done, "done"), (q1, "q1"), (q2, "q2"), (gq1, "gq1"), (gq2, "gq2")):
        _vec(t, nm, b)
    if next_logp is not None:
        _vec(next_logp, "next_logp", b)
        if alpha is None:
            if ent_coef is None:
                raise ValueError("next_logp needs ent_coef or the alpha part")
            _vec(ent_coef, "ent_coef", 1)
    if target_out is not None:
        _vec(target_out, "target_out", b)
    part = None
    if alpha is not None:
        _vec(alpha["logp_pi"], "logp_pi", b)
        for nm in ("log_alpha", "grad_out", "ent_coef_out"):
            _vec(alpha[nm], nm, 1)
        part = nv.AlphaPart(alpha["log_alpha"].data_ptr(), alpha["logp_pi"].data_ptr(), float(alpha["target_entropy"]),
                            alpha["grad_out"].data_ptr(), alpha["ent_coef_out"].data_ptr(),
                            *(None if alpha.get(k) is None else alpha[k].data_ptr() for k in ("loss_out", "loss_sum", "ent_coef_sum")))
    check(nv.lib().cstr_td_twin_q_loss_f32(ptr(q1_t), ptr(q2_t), ptr(next_logp), ptr(rew), ptr(done),
                                           ptr(None if alpha is not None else ent_coef), C.c_float(gamma), ptr(q1), ptr(q2),
                                           C.c_float(scale), ptr(target_out), ptr(gq1), ptr(gq2), ptr(loss_out), ptr(loss_sum),
                                           None if part is None else C.byref(part), C.c_int64(b), stream_ptr()),
          "cstr_td_twin_q_loss_f32")


def sac_actor_loss(logp, q1, q2, ent_coef, g_logp, gq1, gq2, loss_out=None, loss_sum=None):
    b = logp.numel()
    for t, nm in ((logp, "logp"), (q1, "q1"), (q2, "q2"), (g_logp, "g_logp"), (gq1, "gq1"), (gq2, "gq2")):
        _vec(t, nm, b)
    _vec(ent_coef, "ent_coef", 1)
    check(nv.lib().cstr_sac_actor_loss_f32(ptr(logp), ptr(q1), ptr(q2), ptr(ent_coef), ptr(g_logp), ptr(gq1), ptr(gq2),
                                           ptr(loss_out), ptr(loss_sum), C.c_int64(b), stream_ptr()), "cstr_sac_actor_loss_f32")


def neg_mean_loss(q, gq, loss_out=None, loss_sum=None):
    b = q.numel()
    _vec(q, "q", b), _vec(gq, "gq", b)
    check(nv.lib().cstr_neg_mean_loss_f32(ptr(q), ptr(gq), ptr(loss_out), ptr(loss_sum), C.c_int64(b), stream_ptr()),
          "cstr_neg_mean_loss_f32")


# ---- row-chain kernels (csrc/cstr_chain.hip, include/cstr_rl_hip.h "row-chain kernels") -------------------------------------------
def chain_supported(h1: int, h2: int, batch: int) -> bool:
    """Widths multiples of 4; forward launches hold a 16 x (H1 + 4) panel and W1 staged as [H1][8 or 16] in 64 KB of LDS."""
    return (16 <= h1 <= nv.CHAIN_MAX_WIDTH and 16 <= h2 <= nv.CHAIN_MAX_WIDTH and h1 % 4 == 0 and h2 % 4 == 0 and 4 * (1664 + 16 * (h1 + 4) + 16 * h1) <= 65536
            and 16 <= batch <= 1024 and batch % 16 == 0)


def chain_tiles_ok(kdim: int, tiles: int, forward: bool = False) -> bool:
    """A wave holds at most 16 (forward chains: 32) 16-wide chunks of the reduction in registers."""
    s = 4 // tiles
    return -(-((kdim + 15) // 16) // s) <= (32 if forward else 16)


def chain_colgroups(width: int, tiles: int) -> int:
    return (width + 16 * tiles - 1) // (16 * tiles)


def _dp(t):
    return None if t is None else t.data_ptr()


def sac_actor_desc(obs_dim: int, act_dim: int, w1, b1, w2, b2, hw, hb) -> "nv.SacActorNet":
    """hw / hb: the merged [mu | log_std] head ([2A, H2]) or a deterministic actor's last Linear ([A, H2])."""
    h1, h2 = w1.shape[0], w2.shape[0]
    hn = hw.shape[0]
    if hn not in (act_dim, 2 * act_dim):
        raise ValueError(f"head: {hn} outputs for {act_dim} actions")
    for t, nm, shape in ((w1, "w1", (h1, obs_dim)), (b1, "b1", (h1,)), (w2, "w2", (h2, h1)), (b2, "b2", (h2,)), (hw, "hw", (hn, h2)),
                         (hb, "hb", (hn,))):
        _chk(t, nm, shape, th.float32)
    return nv.SacActorNet(obs_dim, act_dim, h1, h2, w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr(), hw.data_ptr(), hb.data_ptr())


def sac_actor_chain_fwd(actor: "nv.SacActorNet", batch: int, x_data, x_pi, x_next, out_done, out_rew, a_h1, a_h2, head_part, tiles: int,
                        ring: Optional[DeviceRing] = None, sample_idx=None, advance_ring: bool = False, head_rng_ctl=None, head_rng_offset: int = 0,
                        eps_all=None, rows_mode: int = 0, head_n: Optional[int] = None):
    """cstr_sac_actor_chain_fwd_f32: gather (or packed observation columns) + layer 1 + layer 2 + head partials of an actor pass
    (rows_mode: CHAIN_ROWS_PAIR = SAC's 2B rows, CHAIN_ROWS_NEXT / CHAIN_ROWS_OBS = B rows of a deterministic actor)."""
    w = actor.obs_dim + actor.act_dim
    ncg = chain_colgroups(actor.h2, tiles)
    head_n = 2 * actor.act_dim if head_n is None else head_n
    m = 2 * batch if rows_mode == nv.CHAIN_ROWS_PAIR else batch
    for t, nm in ((x_pi, "x_pi"), (x_next, "x_next")):
        if t is not None and not (t.is_cuda and t.dtype == th.float32 and tuple(t.shape) == (batch, w) and t.stride() == (w, 1)):
            raise ValueError(f"{nm}: needs a float32 device matrix [{batch}, {w}] with row stride {w}")
    _opt(a_h1, "a_h1", (batch, actor.h1), th.float32), _opt(a_h2, "a_h2", (batch, actor.h2), th.float32)
    _chk(head_part, "head_part", (ncg, m, head_n), th.float32)
    _opt(eps_all, "eps_all", (m, actor.act_dim), th.float32), _opt(head_rng_ctl, "head_rng_ctl", (nv.RNG_CTL_WORDS,), th.int64)
    if not chain_tiles_ok(actor.h1, tiles, forward=True):
        raise ValueError(f"tiles = {tiles}: a wave's share of K = {actor.h1} does not fit")
    if sample_idx is not None:
        _chk(sample_idx, "sample_idx", (2, batch), th.int32)
        _chk(x_data, "x_data", (batch, w), th.float32), _chk(out_done, "out_done", (batch, 1), th.float32), _chk(out_rew, "out_rew", (batch, 1), th.float32)
    check(nv.lib().cstr_sac_actor_chain_fwd_f32(C.byref(actor), None if ring is None else C.byref(ring.c), None if ring is None else ptr(ring.ctl),
                                                C.c_int(1 if advance_ring else 0), ptr(sample_idx),
                                                C.c_int64(batch), ptr(x_data), ptr(x_pi), ptr(x_next), ptr(out_done), ptr(out_rew), ptr(a_h1),
                                                ptr(a_h2), ptr(head_part), ptr(head_rng_ctl), C.c_uint64(int(head_rng_offset)), ptr(eps_all),
                                                C.c_int(rows_mode), C.c_int(head_n), C.c_int(tiles), stream_ptr()),
          "cstr_sac_actor_chain_fwd_f32")


def chain_net(layers, x=None, h1=None, h2=None, q_part=None, role: int = 0) -> "nv.ChainNet":
    """layers = ((w1, b1), (w2, b2), (w3, b3)) of one Q network."""
    (w1, b1), (w2, b2), (w3, b3) = layers
    return nv.ChainNet(w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr(), w3.data_ptr(), b3.data_ptr(), _dp(x), _dp(h1), _dp(h2),
                       _dp(q_part), role, 0)


def q_chain_fwd(nets, w_in: int, obs_dim: int, h1: int, h2: int, batch: int, tiles: int, fin: Optional["nv.SacHeadFin"] = None):
    """cstr_q_chain_fwd_f32: n_nets Q networks (layer 1 recomputed, layer 2 one MFMA column group per workgroup, head as partials)."""
    arr = (nv.ChainNet * len(nets))(*nets)
    check(nv.lib().cstr_q_chain_fwd_f32(arr, C.c_int(len(nets)), C.c_int(w_in), C.c_int(obs_dim), C.c_int(h1), C.c_int(h2), C.c_int64(batch),
                                        None if fin is None else C.byref(fin), C.c_int(tiles), stream_ptr()), "cstr_q_chain_fwd_f32")


def linear_bwd_weight_adam_sets(sets, opts, flat=()):
    """cstr_linear_bwd_weight_adam_sets_f32: dW / db of several Linears AND their Adam steps in one launch. `sets`: [(dz [M, N], x [M, K]
    (row-strided ok), weight parameter, bias parameter, optimiser index, shadow tensor or None[, (weight target, bias target, tau)])];
    the parameters' .grad and moment views
    are looked up in `opts` = [FlatAdam, ...] (pre-advanced control words: see chain_root's adam_advance). `flat`: adam_multi segments
    (no shadow) for parameters without a tile and soft target updates."""
    sets, opts = list(sets), list(opts)
    arr = (nv.WgradAdamSet * len(sets))()
    oarr = (nv.AdamOpt * len(opts))()
    for i, o in enumerate(opts):
        g = o.param_groups[0]
        oarr[i] = nv.AdamOpt(o.ctl.data_ptr(), o.lr_dev.data_ptr(), g["betas"][0], g["betas"][1], g["eps"], o.grad_scale, 0)
    for i, st in enumerate(sets):
        dz, x, weight, bias, oi, shadow = st[:6]
        own = st[6] if len(st) > 6 and st[6] is not None else (None, None, 0.0)
        m, n = dz.shape
        k = x.shape[-1]
        _f32c(dz, f"dz[{i}]")
        if not (x.is_cuda and x.dtype == th.float32 and x.dim() == 2 and x.stride(1) == 1 and x.shape[0] == m):
            raise ValueError(f"x[{i}]: needs a float32 device matrix with unit inner stride and as many rows as dz")
        # a parameter of the optimiser's arena, or an explicit (values, gradient, exp_avg, exp_avg_sq) quadruple of views (merged heads)
        wq = weight if isinstance(weight, tuple) else (weight.detach(), weight.grad, *opts[oi].moments_of(weight))
        bq = bias if isinstance(bias, tuple) else (bias.detach(), bias.grad, *opts[oi].moments_of(bias))
        for t in wq:
            if t is None or _f32c(t, f"weight[{i}]").numel() != n * k:
                raise ValueError(f"set {i}: weight views do not match dz {tuple(dz.shape)} and x {tuple(x.shape)}")
        for t in bq:
            if t is None or _f32c(t, f"bias[{i}]").numel() != n:
                raise ValueError(f"set {i}: bias views do not match dz {tuple(dz.shape)}")
        if shadow is not None and _f32c(shadow, "shadow").numel() != swizzled_numel(n, k):
            raise ValueError("shadow: wrong size")
        arr[i] = nv.WgradAdamSet(nv.WgradSet(dz.data_ptr(), x.data_ptr(), max(x.stride(0), k), wq[1].data_ptr(), bq[1].data_ptr(), m, n, k),
                                 wq[0].data_ptr(), wq[2].data_ptr(), wq[3].data_ptr(), bq[0].data_ptr(), bq[2].data_ptr(), bq[3].data_ptr(),
                                 _dp(shadow), oi, 0, _dp(own[0]), _dp(own[1]), float(own[2]), 0.0)
        if own[0] is not None and (_f32c(own[0], "weight target").numel() != n * k or _f32c(own[1], "bias target").numel() != n):
            raise ValueError(f"set {i}: target views do not match the parameters")
    farr, nf = _adam_segments(flat)
    check(nv.lib().cstr_linear_bwd_weight_adam_sets_f32(arr, C.c_int(len(sets)), oarr, C.c_int(len(opts)), farr, C.c_int(nf), stream_ptr()),
          "cstr_linear_bwd_weight_adam_sets_f32")


def chain_root(mode: str, batch: int, q_parts, b3s, n_parts: int, gamma: float = 0.0, scale: float = 0.0, next_logp=None, rew=None, done=None,
               ent_coef=None, logp=None, target_out=None, q_out=None, gq_out=None, loss_out=None, loss_sum=None, alpha: Optional[dict] = None,
               rng_advance=None, adam_advance=()) -> "nv.ChainRoot":
    part = nv.AlphaPart()
    if alpha is not None:
        part = nv.AlphaPart(alpha["log_alpha"].data_ptr(), alpha["logp_pi"].data_ptr(), float(alpha["target_entropy"]), alpha["grad_out"].data_ptr(),
                            alpha["ent_coef_out"].data_ptr(), *(_dp(alpha.get(k)) for k in ("loss_out", "loss_sum", "ent_coef_sum")))
    qp, bb = (C.c_void_p * 4)(), (C.c_void_p * 4)()
    for i, (q, b) in enumerate(zip(q_parts, b3s)):
        qp[i], bb[i] = q.data_ptr(), b.data_ptr()
    rc_ptr, adv = (None, 0) if rng_advance is None else (rng_advance[0].data_ptr(), int(rng_advance[1]))
    actl, ab1, ab2 = (C.c_void_p * 2)(), (C.c_double * 2)(), (C.c_double * 2)()
    for i, opt in enumerate(adam_advance):  # FlatAdam optimisers whose step counter this launch advances (<= 2)
        actl[i], (ab1[i], ab2[i]) = opt.ctl.data_ptr(), opt.param_groups[0]["betas"]
    return nv.ChainRoot({"td": 1, "sac_actor": 2, "neg_mean": 3}[mode], batch, float(gamma), float(scale), qp, bb, n_parts, 0, _dp(next_logp), _dp(rew),
                        _dp(done), None if alpha is not None else _dp(ent_coef), _dp(logp), _dp(target_out), _dp(q_out), _dp(gq_out), _dp(loss_out),
                        _dp(loss_sum), part, rc_ptr, adv, actl, ab1, ab2)


def q_chain_bwd(nets, root: "nv.ChainRoot", w_in: int, obs_dim: int, h1: int, h2: int, tiles: int, dz2=None, dz1=None, gact_part=None):
    """cstr_q_chain_bwd_f32: loss root + dz2 (recomputed) + one column group of dz1 (+ partial action gradients)."""
    arr = (nv.ChainNet * len(nets))(*nets)
    check(nv.lib().cstr_q_chain_bwd_f32(arr, C.c_int(len(nets)), C.byref(root), C.c_int(w_in), C.c_int(obs_dim), C.c_int(h1), C.c_int(h2), ptr(dz2),
                                        ptr(dz1), ptr(gact_part), C.c_int(tiles), stream_ptr()), "cstr_q_chain_bwd_f32")


def sac_actor_chain_bwd(actor: "nv.SacActorNet", gact_part, n_nets: int, n_parts: int, ent_coef, x_pi, params, eps, a_h1, a_h2, g_params, dz2, dz1,
                        batch: int, tiles: int, kind: int = 0):
    """cstr_sac_actor_chain_bwd_f32: action gradient from the critic's partials, head backward (kind: CHAIN_HEAD_GAUSSIAN / _DETERMINISTIC),
    dz2 (recomputed), dz1 column group."""
    hn = actor.act_dim if kind == nv.CHAIN_HEAD_DETERMINISTIC else 2 * actor.act_dim
    _chk(gact_part, "gact_part", (n_nets, n_parts, batch, actor.act_dim), th.float32)
    _chk(g_params, "g_params", (batch, hn), th.float32), _chk(dz2, "dz2", (batch, actor.h2), th.float32)
    _chk(dz1, "dz1", (batch, actor.h1), th.float32)
    check(nv.lib().cstr_sac_actor_chain_bwd_f32(C.byref(actor), ptr(gact_part), C.c_int(n_nets), C.c_int(n_parts), ptr(ent_coef), ptr(x_pi), ptr(params),
                                                ptr(eps), ptr(a_h1), ptr(a_h2), ptr(g_params), ptr(dz2), ptr(dz1), C.c_int64(batch), C.c_int(kind),
                                                C.c_int(tiles), stream_ptr()), "cstr_sac_actor_chain_bwd_f32")


def chain_sum_parts(part, out):
    """cstr_chain_sum_parts_f32: out [rows, cols] (row-strided ok) = sum over the leading dimension(s) of part [..., rows, cols]."""
    rows, cols = out.shape
    n_parts = part.numel() // (rows * cols)
    if part.numel() != n_parts * rows * cols or _f32c(part, "part") is None:
        raise ValueError("part: needs [n_parts, rows, cols]")
    if not (out.is_cuda and out.dtype == th.float32 and out.stride(1) == 1):
        raise ValueError("out: needs a float32 device matrix with unit inner stride")
    check(nv.lib().cstr_chain_sum_parts_f32(ptr(part), C.c_int(n_parts), C.c_int64(rows), C.c_int(cols), ptr(out), C.c_int64(out.stride(0)), stream_ptr()),
          "cstr_chain_sum_parts_f32")
