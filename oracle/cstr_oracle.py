"""ctypes bindings for oracle/cstr_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The product package never does (tests/test_layout.py greps for it).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libcstr_oracle.so")


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "cstr_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libcstr_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


class Coef(C.Structure):
    _fields_ = [(n, C.c_float) for n in (
        "q_v1", "q_v2", "cf", "tf", "tcf", "k0", "neg_e", "r_gas", "hk", "rho_cp", "cool1", "cool2", "neg_ua1",
        "neg_ua2", "rho_c", "c_pc", "dt")] + [
        ("s_lo", C.c_float * 4), ("s_hi", C.c_float * 4), ("s_span", C.c_float * 4),
        ("a_lo", C.c_float * 2), ("a_hi", C.c_float * 2), ("a_span", C.c_float * 2),
        ("target_c2", C.c_float), ("conc_span", C.c_float), ("max_steps", C.c_int32)]


class MTState(C.Structure):
    _fields_ = [("key", C.c_uint32 * 624), ("pos", C.c_int32), ("has_gauss", C.c_int32), ("gauss", C.c_double)]


class Ring(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("obs", "next_obs", "act", "rew", "done", "timeout")] + [
        ("rows", C.c_int64), ("n_envs", C.c_int64), ("pos", C.c_int64),
        ("obs_dim", C.c_int32), ("act_dim", C.c_int32), ("full", C.c_int32)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.mt19937_randint_cpu.restype = C.c_int64
        _lib.replay_sample_mt19937_cpu.restype = C.c_int64
        _lib.cstr_collect_loop_cpu.restype = C.c_int64
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def default_coef(target_c2=0.20, min_conc=0.05, max_conc=0.45, max_steps=400) -> Coef:
    c = Coef()
    lib().cstr_default_coef(C.byref(c), C.c_double(target_c2), C.c_double(min_conc), C.c_double(max_conc), C.c_int32(max_steps))
    return c


def vec_step(obs, act, step_count, reset_obs=None, integrator="euler", coef=None, n_threads=1):
    """Batched VecEnv.step with auto-reset. Returns (next_obs, obs_after, reward, done, timeout, step_count_after)."""
    obs, act = _f32(obs), _f32(act)
    n = obs.shape[0]
    assert obs.shape == (n, 4) and act.shape == (n, 2)
    steps = np.ascontiguousarray(step_count, dtype=np.int32).copy()
    reset_obs = obs.copy() if reset_obs is None else _f32(reset_obs)
    coef = coef or default_coef()
    nxt, after = np.empty_like(obs), np.empty_like(obs)
    rew, done, tout = (np.empty(n, np.float32) for _ in range(3))
    lib().cstr_vec_step_f32_cpu(C.byref(coef), C.c_int({"euler": 0, "rk4": 1}[integrator]), _p(obs), _p(act), _p(steps),
                                _p(reset_obs), _p(nxt), _p(after), _p(rew), _p(done), _p(tout), C.c_int64(n), C.c_int(n_threads))
    return nxt, after, rew, done, tout, steps


def action_scale_chain(a, squashed, low, high):
    a = _f32(a)
    low, high = _f32(low), _f32(high)
    buf, env = np.empty_like(a), np.empty_like(a)
    lib().action_scale_chain_f32_cpu(_p(a), C.c_int(int(squashed)), _p(low), _p(high), _p(buf), _p(env),
                                     C.c_int64(a.shape[0]), C.c_int(a.shape[1]))
    return buf, env


class MT19937:
    """np.random.seed(seed) / np.random.randint(0, high, size) restated."""

    def __init__(self, seed: int):
        self.st = MTState()
        lib().mt19937_seed_cpu(C.byref(self.st), C.c_uint32(seed & 0xFFFFFFFF))

    def randint(self, high: int, size: int) -> np.ndarray:
        out = np.empty(size, np.int64)
        self.last_used = lib().mt19937_randint_cpu(C.byref(self.st), C.c_int64(high), _p(out), C.c_int64(size))
        return out

    def normal(self, loc, scale, n_calls: int = 1) -> np.ndarray:
        """`n_calls` consecutive np.random.normal(loc, scale).astype(float32) -> [n_calls, len(loc)]"""
        loc, scale = np.ascontiguousarray(loc, np.float64).ravel(), np.ascontiguousarray(scale, np.float64).ravel()
        out = np.empty((n_calls, loc.size), np.float32)
        lib().mt19937_normal_f32_cpu(C.byref(self.st), _p(loc), _p(scale), C.c_int(loc.size), _p(out), C.c_int64(out.size))
        return out

    def words(self) -> np.ndarray:
        """the 628-word device image {key[624], pos, has_gauss, gauss f64}"""
        return np.frombuffer(bytes(self.st), dtype=np.uint32).copy()

    @property
    def key(self):
        return np.frombuffer(self.st.key, dtype=np.uint32).copy()

    @property
    def pos(self):
        return int(self.st.pos)

    def set_state(self, key, pos):
        C.memmove(self.st.key, np.ascontiguousarray(key, np.uint32).ctypes.data, 624 * 4)
        self.st.pos = int(pos)


class ReplayRing:
    """ReplayBuffer restated (fields [R][N][D]/[R][N][A]/[R][N], all f32)."""

    def __init__(self, rows, n_envs, obs_dim, act_dim):
        self.observations = np.zeros((rows, n_envs, obs_dim), np.float32)
        self.next_observations = np.zeros((rows, n_envs, obs_dim), np.float32)
        self.actions = np.zeros((rows, n_envs, act_dim), np.float32)
        self.rewards = np.zeros((rows, n_envs), np.float32)
        self.dones = np.zeros((rows, n_envs), np.float32)
        self.timeouts = np.zeros((rows, n_envs), np.float32)
        self.c = Ring(_p(self.observations).value, _p(self.next_observations).value, _p(self.actions).value,
                      _p(self.rewards).value, _p(self.dones).value, _p(self.timeouts).value,
                      rows, n_envs, 0, obs_dim, act_dim, 0)

    pos = property(lambda self: int(self.c.pos))
    full = property(lambda self: bool(self.c.full))

    def add(self, obs, next_obs, act, rew, done, timeout):
        a = [_f32(x) for x in (obs, next_obs, act, rew, done, timeout)]
        lib().replay_add_f32_cpu(C.byref(self.c), *[_p(x) for x in a])

    def sample(self, mt: MT19937, batch: int):
        D, A = self.c.obs_dim, self.c.act_dim
        o, a, no = np.empty((batch, D), np.float32), np.empty((batch, A), np.float32), np.empty((batch, D), np.float32)
        d, r = np.empty((batch, 1), np.float32), np.empty((batch, 1), np.float32)
        bi, ei = np.empty(batch, np.int64), np.empty(batch, np.int64)
        used = lib().replay_sample_mt19937_cpu(C.byref(self.c), C.byref(mt.st), C.c_int64(batch), _p(o), _p(a), _p(no),
                                               _p(d), _p(r), _p(bi), _p(ei))
        if used < 0:
            raise ValueError("high <= 0")
        return (o, a, no, d, r), (bi, ei)


def td_target_min(q1, q2, logp, rew, done, alpha, gamma):
    q1, q2, rew, done = (_f32(x).reshape(-1) for x in (q1, q2, rew, done))
    lp = None if logp is None else _f32(logp).reshape(-1)
    out = np.empty_like(q1)
    lib().td_target_min_f32_cpu(_p(q1), _p(q2), _p(lp), _p(rew), _p(done), C.c_float(alpha), C.c_float(gamma), _p(out),
                                C.c_int64(q1.size))
    return out


def polyak(param, target, tau):
    param, target = _f32(param).reshape(-1), _f32(target).reshape(-1).copy()
    lib().polyak_f32_cpu(_p(param), _p(target), C.c_double(tau), C.c_int64(param.size))
    return target


def adam_step(param, grad, exp_avg, exp_avg_sq, step, lr, beta1=0.9, beta2=0.999, eps=1e-8):
    p, m, v = (_f32(x).reshape(-1).copy() for x in (param, exp_avg, exp_avg_sq))
    g = _f32(grad).reshape(-1)
    lib().adam_f32_cpu(_p(p), _p(g), _p(m), _p(v), C.c_int64(step), C.c_double(lr), C.c_double(beta1), C.c_double(beta2),
                       C.c_double(eps), C.c_int64(p.size))
    return p, m, v


PCG_DTYPE = np.dtype([("state_hi", "<u8"), ("state_lo", "<u8"), ("inc_hi", "<u8"), ("inc_lo", "<u8")])


def pcg64_states_from_seeds(seeds) -> np.ndarray:
    """Host-side seeding via numpy's own SeedSequence/PCG64 (never restated)."""
    out = np.zeros(len(seeds), PCG_DTYPE)
    m = (1 << 64) - 1
    for i, s in enumerate(seeds):
        st = np.random.PCG64(np.random.SeedSequence(int(s))).state["state"]
        out[i] = (st["state"] >> 64, st["state"] & m, st["inc"] >> 64, st["inc"] & m)
    return out


STATIC_INIT_STATE = (0.45, 310.0, 0.25, 290.0)  # twoseriescstr.py:96


def reset_draw(pcg_states: np.ndarray, mask=None, static_init=None) -> np.ndarray:
    """static_init: f64 [n, 4] = every env's `init_state` (init_mode="static"), updated in place; None = "random"."""
    n = len(pcg_states)
    obs = np.zeros((n, 4), np.float32)
    mk = None if mask is None else np.ascontiguousarray(mask, np.uint8)
    if static_init is not None:
        assert static_init.dtype == np.float64 and static_init.shape == (n, 4) and static_init.flags.c_contiguous
    lib().cstr_reset_draw_batch_cpu(_p(pcg_states), _p(mk), _p(static_init), _p(obs), C.c_int64(n))
    return obs


def collect_loop(ring: ReplayRing, obs, act, step_count, reset_obs, n_steps, n_threads, coef=None):
    """cpu_baseline leg of bench.py: n_steps x (vec step + ring add). Mutates obs/step_count in place."""
    coef = coef or default_coef()
    n = obs.shape[0]
    scratch = np.empty(n * 7, np.float32)
    return lib().cstr_collect_loop_cpu(C.byref(coef), C.byref(ring.c), _p(obs), _p(act), _p(step_count), _p(reset_obs),
                                       _p(scratch), C.c_int64(n_steps), C.c_int(n_threads))
