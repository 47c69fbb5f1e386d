"""ctypes binding of libcstr_rl_hip.so (the C ABI declared in include/cstr_rl_hip.h).

The library is the product's only implementation of the env / replay / element-wise hot path:
there is NO CPU or PyTorch fallback. `lib()` raises if the shared object is missing or a symbol
declared in the header is not exported.
"""
import ctypes as C
import os
from typing import Optional

_PKG_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB_PATH = os.environ.get("CSTR_LIB_PATH", os.path.join(_PKG_ROOT, "libcstr_rl_hip.so"))  # override: kernel A/B builds

INTEGRATORS = {"euler": 0, "rk4": 1}
RING_CTL_WORDS, ADAM_CTL_WORDS, MT_STATE_WORDS, PCG_STATE_WORDS, MAX_SAMPLE_BATCH = 4, 4, 628, 4, 16384
MAX_NOISE_PERIOD = 8
VECNORM_STATE_WORDS = 20
RNG_CTL_WORDS, MAX_HEAD_ACT, MAX_LINEAR_SETS, MAX_ADAM_SEGS = 16, 4, 16, 4

SYMBOLS = (
    "cstr_abi_version", "cstr_error_string", "cstr_default_coef", "cstr_vec_step_f32", "cstr_reset_draw_f32",
    "cstr_replay_add_f32", "cstr_collect_step_f32", "cstr_collect_step_rng_f32", "cstr_mt19937_seed", "cstr_mt19937_normal_f32", "cstr_mt19937_normal_f64", "cstr_replay_sample_mt19937_f32", "cstr_replay_sample_packed_mt19937_f32", "cstr_replay_gather_packed_f32", "cstr_rollout_step_f32", "cstr_linear_act_fwd_gather_f32",
    "cstr_adam_multi_f32", "cstr_gaussian_head_fwd_f32", "cstr_gaussian_head_gemm_fwd_f32", "cstr_gaussian_head_bwd_f32", "cstr_gaussian_head_bwd_input_f32", "cstr_linear_act_fwd_f32", "cstr_linear_act_fwd_sets_f32", "cstr_linear_bwd_input_f32", "cstr_linear_bwd_weight_f32", "cstr_linear_bwd_weight_sets_f32", "cstr_td_twin_q_loss_f32", "cstr_policy_rows_fwd_f32", "cstr_policy_swizzle_f32", "cstr_target_smooth_f32", "cstr_linear_smooth_fwd_f32", "cstr_hidden_head_fwd_f32", "cstr_hidden_head_bwd_f32", "cstr_hidden_head_bwd_root_f32", "cstr_vecnorm_init_f64", "cstr_vecnorm_step_f64", "cstr_vecnorm_apply_f32",
    "cstr_td_target_min_f32", "cstr_polyak_f32", "cstr_adam_f32", "cstr_bias_act_fwd_f32", "cstr_bias_act_bwd_f32", "cstr_bias_act_bwd_rows_f32",
    "cstr_squashed_gaussian_fwd_f32", "cstr_squashed_gaussian_bwd_f32", "cstr_sac_alpha_f32", "cstr_twin_q_loss_f32",
    "cstr_sac_actor_loss_f32", "cstr_neg_mean_loss_f32",
    "cstr_sac_actor_chain_fwd_f32", "cstr_q_chain_fwd_f32", "cstr_q_chain_bwd_f32", "cstr_sac_actor_chain_bwd_f32",
    "cstr_linear_bwd_weight_adam_sets_f32", "cstr_chain_sum_parts_f32",
)


class Coef(C.Structure):
    """cstr_coef_t"""
    _fields_ = [(n, C.c_float) for n in (
        "q_v1", "q_v2", "cf", "tf", "tcf", "k0", "neg_e", "r_gas", "hk", "rho_cp", "cool1", "cool2", "neg_ua1",
        "neg_ua2", "rho_c", "c_pc", "dt")] + [
        ("s_lo", C.c_float * 4), ("s_hi", C.c_float * 4), ("s_span", C.c_float * 4),
        ("a_lo", C.c_float * 2), ("a_hi", C.c_float * 2), ("a_span", C.c_float * 2),
        ("target_c2", C.c_float), ("conc_span", C.c_float), ("max_steps", C.c_int32)]


class Ring(C.Structure):
    """cstr_ring_t"""
    _fields_ = [(n, C.c_void_p) for n in ("obs", "next_obs", "act", "rew", "done", "timeout")] + [
        ("rows", C.c_int64), ("n_envs", C.c_int64), ("obs_dim", C.c_int32), ("act_dim", C.c_int32)]


class AdamSeg(C.Structure):
    """cstr_adam_seg_t"""
    _fields_ = [("param", C.c_void_p), ("grad", C.c_void_p), ("exp_avg", C.c_void_p), ("exp_avg_sq", C.c_void_p),
                ("adam_ctl", C.c_void_p), ("lr", C.c_void_p), ("beta1", C.c_double), ("beta2", C.c_double), ("eps", C.c_double),
                ("grad_scale", C.c_float), ("n", C.c_int64), ("polyak_source", C.c_void_p), ("tau", C.c_double),
                ("shadow", C.c_void_p), ("shadow_begin", C.c_int64), ("shadow_n", C.c_int64), ("shadow_k", C.c_int64), ("own_target", C.c_void_p)]


class LinearSet(C.Structure):
    """cstr_linear_set_t"""
    _fields_ = [("x", C.c_void_p), ("ldx", C.c_int64), ("w", C.c_void_p), ("bias", C.c_void_p), ("y", C.c_void_p), ("ldy", C.c_int64)]


class WgradSet(C.Structure):
    """cstr_wgrad_set_t"""
    _fields_ = [("dz", C.c_void_p), ("x", C.c_void_p), ("ldx", C.c_int64), ("dw", C.c_void_p), ("db", C.c_void_p),
                ("m", C.c_int64), ("n", C.c_int64), ("k", C.c_int64)]


class AlphaPart(C.Structure):
    """cstr_alpha_part_t"""
    _fields_ = [("log_alpha", C.c_void_p), ("logp_pi", C.c_void_p), ("target_entropy", C.c_float), ("grad_out", C.c_void_p),
                ("ent_coef_out", C.c_void_p), ("loss_out", C.c_void_p), ("loss_sum", C.c_void_p), ("ent_coef_sum", C.c_void_p)]


class HeadRoot(C.Structure):
    """cstr_head_root_t"""
    _fields_ = [("mode", C.c_int32), ("batch", C.c_int32), ("gamma", C.c_float), ("scale", C.c_float), ("q1_t", C.c_void_p),
                ("q2_t", C.c_void_p), ("next_logp", C.c_void_p), ("rew", C.c_void_p), ("done", C.c_void_p), ("ent_coef", C.c_void_p),
                ("q1", C.c_void_p), ("q2", C.c_void_p), ("target_out", C.c_void_p), ("logp", C.c_void_p), ("g_logp", C.c_void_p),
                ("loss_out", C.c_void_p), ("loss_sum", C.c_void_p), ("alpha", AlphaPart)]


class PolicyMlp(C.Structure):
    """cstr_policy_mlp_t"""
    _fields_ = [("k0", C.c_int32), ("h1", C.c_int32), ("h2", C.c_int32), ("act_dim", C.c_int32), ("act", C.c_int32), ("head", C.c_int32),
                ("out_act", C.c_int32), ("reserved", C.c_int32), ("w1", C.c_void_p), ("b1", C.c_void_p), ("w2", C.c_void_p),
                ("b2", C.c_void_p), ("w3", C.c_void_p), ("b3", C.c_void_p), ("w2_swizzled", C.c_void_p)]


CHAIN_MAX_NETS, CHAIN_MAX_WIDTH = 16, 512
CHAIN_ROLE_PLAIN, CHAIN_ROLE_STORE_PI, CHAIN_ROLE_NEXT, CHAIN_ROLE_NEXT_STORE, CHAIN_ROLE_PI = 0, 1, 2, 3, 4
CHAIN_ROWS_PAIR, CHAIN_ROWS_NEXT, CHAIN_ROWS_OBS = 0, 1, 2
CHAIN_HEAD_GAUSSIAN, CHAIN_HEAD_DETERMINISTIC = 0, 1


class ChainNet(C.Structure):
    """cstr_chain_net_t"""
    _fields_ = [(n, C.c_void_p) for n in ("w1", "b1", "w2", "b2", "w3", "b3", "x", "h1", "h2", "q_part")] + [("role", C.c_int32), ("reserved", C.c_int32)]


class SacActorNet(C.Structure):
    """cstr_sac_actor_t"""
    _fields_ = [(n, C.c_int32) for n in ("obs_dim", "act_dim", "h1", "h2")] + [(n, C.c_void_p) for n in ("w1", "b1", "w2", "b2", "hw", "hb")]


class SacHeadFin(C.Structure):
    """cstr_sac_head_fin_t"""
    _fields_ = [(n, C.c_void_p) for n in ("head_part", "hb", "eps")] + [(n, C.c_int32) for n in ("n_parts", "act_dim", "obs_dim", "kind", "part_rows",
                                                                                                 "next_offset")] + [
        ("sigma", C.c_float), ("clip", C.c_float)] + [(n, C.c_void_p) for n in ("x_pi", "x_next", "params", "logp_pi", "logp_next")]


class ChainRoot(C.Structure):
    """cstr_chain_root_t"""
    _fields_ = [("mode", C.c_int32), ("batch", C.c_int32), ("gamma", C.c_float), ("scale", C.c_float), ("q_part", C.c_void_p * 4),
                ("b3", C.c_void_p * 4), ("n_parts", C.c_int32), ("reserved", C.c_int32)] + [
        (n, C.c_void_p) for n in ("next_logp", "rew", "done", "ent_coef", "logp", "target_out", "q_out", "gq_out", "loss_out", "loss_sum")] + [
        ("alpha", AlphaPart), ("rng_ctl", C.c_void_p), ("rng_advance", C.c_uint64), ("adam_advance", C.c_void_p * 2),
        ("adam_beta1", C.c_double * 2), ("adam_beta2", C.c_double * 2)]


class AdamOpt(C.Structure):
    """cstr_adam_opt_t"""
    _fields_ = [("adam_ctl", C.c_void_p), ("lr", C.c_void_p), ("beta1", C.c_double), ("beta2", C.c_double), ("eps", C.c_double),
                ("grad_scale", C.c_float), ("reserved", C.c_int32)]


class WgradAdamSet(C.Structure):
    """cstr_wgrad_adam_set_t"""
    _fields_ = [("g", WgradSet)] + [(n, C.c_void_p) for n in ("w", "w_m", "w_v", "b", "b_m", "b_v", "shadow")] + [("opt", C.c_int32), ("reserved", C.c_int32),
                                                                                                             ("w_target", C.c_void_p), ("b_target", C.c_void_p),
                                                                                                             ("tau", C.c_float), ("reserved2", C.c_float)]


class VecNormCfg(C.Structure):
    """cstr_vecnorm_cfg_t"""
    _fields_ = [("training", C.c_int32), ("norm_obs", C.c_int32), ("norm_reward", C.c_int32), ("obs_dim", C.c_int32),
                ("clip_obs", C.c_double), ("clip_reward", C.c_double), ("gamma", C.c_double), ("epsilon", C.c_double)]


class NativeError(RuntimeError):
    pass


_lib: Optional[C.CDLL] = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        # PyTorch ships its own HIP runtime: it has to be the FIRST one loaded into the process. Loading this library (linked
        # against /opt/rocm's libamdhip64) before `import torch` leaves two runtimes, and launches on torch's memory through the
        # other one fail with "no ROCm-capable device is detected" (seen with `python __graft_entry__.py smoke`: build() loaded the
        # library, then smoke() imported torch).
        import torch  # noqa: F401

        if not os.path.exists(LIB_PATH):
            raise NativeError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(or `make -C pytorch-rl-enhancedstablebaselines_amd/csrc`). This stack has no CPU/PyTorch fallback.")
        l = C.CDLL(LIB_PATH)
        missing = [s for s in SYMBOLS if not hasattr(l, s)]
        if missing:
            raise NativeError(f"{LIB_PATH} does not export {missing}")
        l.cstr_error_string.restype = C.c_char_p
        l.cstr_default_coef.restype = None
        if l.cstr_abi_version() != 5:
            raise NativeError("libcstr_rl_hip.so ABI version mismatch")
        _lib = l
    return _lib


ABI_CALLS = [0]  # launching entry points called so far (every one of them reports through check()): launch census of a recorded graph


def check(rc: int, what: str) -> None:
    """Error convention of the ABI: 0 ok, <0 cstr error, >0 hipError_t -> RuntimeError."""
    ABI_CALLS[0] += 1
    if rc != 0:
        raise NativeError(f"{what} failed: {lib().cstr_error_string(C.c_int(rc)).decode()} (code {rc})")


def default_coef(target_c2=0.20, min_conc=0.05, max_conc=0.45, max_steps=400) -> Coef:
    c = Coef()
    lib().cstr_default_coef(C.byref(c), C.c_double(target_c2), C.c_double(min_conc), C.c_double(max_conc),
                            C.c_int32(max_steps))
    return c


def ptr(t) -> C.c_void_p:
    """Raw device pointer of a torch tensor (None -> NULL)."""
    return C.c_void_p(None if t is None else t.data_ptr())


def stream_ptr() -> C.c_void_p:
    import torch

    return C.c_void_p(torch.cuda.current_stream().cuda_stream)
