"""CPU-side checks of the C-ABI library: it loads, exports every symbol include/cstr_rl_hip.h
declares, folds the CSTR constants exactly like the oracle, and rejects bad arguments without
touching a GPU (no compute calls here)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import ROOT
from core import _native as nv
from oracle import cstr_oracle as orc


def _header_functions():
    src = open(os.path.join(ROOT, "include", "cstr_rl_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(cstr_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    lib = nv.lib()
    declared = _header_functions()
    assert len(declared) >= 12
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/cstr_rl_hip.h but not exported"
    assert sorted(nv.SYMBOLS) == declared
    assert lib.cstr_abi_version() == 5


def test_error_strings():
    lib = nv.lib()
    assert lib.cstr_error_string(0) == b"ok"
    assert b"bad argument" in lib.cstr_error_string(-1)
    assert b"unsupported" in lib.cstr_error_string(-2)


def test_coef_folding_matches_oracle():
    a, b = nv.default_coef(0.2, 0.05, 0.45, 400), orc.default_coef(0.2, 0.05, 0.45, 400)
    assert C.sizeof(a) == C.sizeof(b)
    assert bytes(a) == bytes(b)
    # NEP-50 folding spot checks (twoseriescstr.py:37-61)
    assert a.hk == np.float32(6.78e4 * 7.2e10) and a.cool1 == np.float32((1000 * 0.239) / (1000 * 0.239 * 100))
    assert a.s_span[1] == np.float32(400.0) - np.float32(273.15) and a.conc_span == np.float32(0.45 - 0.05)


def test_bad_arguments_are_rejected_on_the_host():
    lib = nv.lib()
    coef = nv.default_coef()
    null = C.c_void_p(None)
    assert lib.cstr_vec_step_f32(C.byref(coef), 0, 4, 2, null, null, null, null, null, null, null, null, null, C.c_int64(8), null) == -1
    assert lib.cstr_polyak_f32(null, null, C.c_double(0.005), C.c_int64(4), null) == -1
    assert lib.cstr_td_target_min_f32(null, null, null, null, null, null, C.c_float(0.99), null, C.c_int64(4), null) == -1
    fake = C.c_void_p(0x1000)  # never dereferenced: argument checks fail first
    assert lib.cstr_vec_step_f32(C.byref(coef), 0, 5, 2, fake, fake, fake, fake, fake, fake, fake, fake, fake, C.c_int64(8), null) == -2
    assert lib.cstr_vec_step_f32(C.byref(coef), 0, 4, 4, fake, fake, fake, fake, fake, fake, fake, fake, fake, C.c_int64(8), null) == -2  # (4,4) is not a layout
    assert lib.cstr_vec_step_f32(C.byref(coef), 7, 4, 2, fake, fake, fake, fake, fake, fake, fake, fake, fake, C.c_int64(8), null) == -2
    ring = nv.Ring(0x1000, 0x1000, 0x1000, 0x1000, 0x1000, 0x1000, 4, 4, 4, 3)  # act_dim 3
    assert lib.cstr_replay_sample_mt19937_f32(C.byref(ring), fake, fake, C.c_int64(8), fake, fake, fake, fake, fake, null, null, null) == -2
    ring = nv.Ring(0x1000, 0x1000, 0x1000, 0x1000, 0x1000, 0x1000, 4, 4, 4, 2)
    assert lib.cstr_replay_sample_mt19937_f32(C.byref(ring), fake, fake, C.c_int64(1 << 20), fake, fake, fake, fake, fake, null, null, null) == -2
    misaligned = C.c_void_p(0x1004)
    assert lib.cstr_polyak_f32(misaligned, fake, C.c_double(0.005), C.c_int64(4), null) == -1


def test_hip_ops_refuse_cpu_tensors():
    import torch as th

    from core.common import hip_ops

    with pytest.raises(ValueError, match="No CPU fallback"):
        hip_ops.polyak(th.zeros(8), th.zeros(8), 0.005)


def test_round2_entry_points_reject_bad_arguments_on_the_host():
    """cstr_hidden_head_bwd_root_f32, cstr_bias_act_bwd_rows_f32, cstr_collect_step_rng_f32 and the policy launch's flag word:
    argument checks fail before anything is dereferenced or launched."""
    lib = nv.lib()
    null, fake = C.c_void_p(None), C.c_void_p(0x1000)
    i64, f32 = C.c_int64, C.c_float
    # loss root: NULL struct, unknown mode, batch != m, missing mode-1 / mode-2 pointers, too many rows
    assert lib.cstr_hidden_head_bwd_root_f32(null, fake, 1, fake, fake, null, null, null, i64(256), i64(256), null) == -1
    rt = nv.HeadRoot(7, 256, 0.99, 1.0, 0x1000, 0x1000, None, 0x1000, 0x1000, None, 0x1000, 0x1000, None, None, None, None, None, nv.AlphaPart())
    assert lib.cstr_hidden_head_bwd_root_f32(C.byref(rt), fake, 1, fake, fake, null, null, null, i64(256), i64(256), null) == -1
    rt.mode, rt.batch = 1, 128
    assert lib.cstr_hidden_head_bwd_root_f32(C.byref(rt), fake, 1, fake, fake, null, null, null, i64(256), i64(256), null) == -1
    rt.batch, rt.rew = 256, None
    assert lib.cstr_hidden_head_bwd_root_f32(C.byref(rt), fake, 1, fake, fake, null, null, null, i64(256), i64(256), null) == -1
    rt.rew, rt.mode = 0x1000, 2  # mode 2 needs logp, g_logp, ent_coef
    assert lib.cstr_hidden_head_bwd_root_f32(C.byref(rt), fake, 1, fake, fake, null, null, null, i64(256), i64(256), null) == -1
    rt.mode, rt.q1 = 3, None  # mode 3 (-mean(Q1)) still needs q1
    assert lib.cstr_hidden_head_bwd_root_f32(C.byref(rt), fake, 1, fake, fake, null, null, null, i64(256), i64(256), null) == -1
    rt.q1 = 0x1000
    rt.mode, rt.batch = 1, 2048
    assert lib.cstr_hidden_head_bwd_root_f32(C.byref(rt), fake, 1, fake, fake, null, null, null, i64(2048), i64(256), null) == -2
    rt.batch = 256  # gb1 / gw2 / gb2 go together
    assert lib.cstr_hidden_head_bwd_root_f32(C.byref(rt), fake, 1, fake, fake, fake, null, null, i64(256), i64(256), null) == -1
    # strided bias / activation backward: row stride below the width, missing y for an activation, in-place with a strided gy
    assert lib.cstr_bias_act_bwd_rows_f32(fake, i64(3), fake, i64(4), 2, fake, null, i64(8), i64(4), null) == -1
    assert lib.cstr_bias_act_bwd_rows_f32(fake, i64(6), null, i64(6), 2, fake, null, i64(8), i64(4), null) == -1
    assert lib.cstr_bias_act_bwd_rows_f32(fake, i64(6), fake, i64(6), 0, fake, null, i64(8), i64(4), null) == -1
    assert lib.cstr_bias_act_bwd_rows_f32(fake, i64(6), fake, i64(6), 5, C.c_void_p(0x2000), null, i64(8), i64(4), null) == -2
    # the collect step's rng form shares the checks of the plain one
    coef = nv.default_coef()
    ring = nv.Ring(0x1000, 0x1000, 0x1000, 0x1000, 0x1000, 0x1000, 4, 8, 4, 2)
    lo = (C.c_float * 2)(-1, -1)
    hi = (C.c_float * 2)(1, 1)
    assert lib.cstr_collect_step_rng_f32(C.byref(coef), 0, C.byref(ring), fake, fake, fake, fake, 1, lo, hi, null, null, null, null, null, null,
                                         null, null, fake, C.c_uint64(8), null) == -1  # no reset source
    # the policy launch's flag word: only bit 0 is defined
    net = nv.PolicyMlp(4, 64, 64, 2, 1, 0, 0, 2, 0x1000, 0x1000, 0x1000, 0x1000, 0x1000, 0x1000, None)
    assert lib.cstr_policy_rows_fwd_f32(C.byref(net), fake, i64(4), null, fake, fake, i64(2), null, i64(16), null) == -1


def test_rollout_entry_points_reject_bad_arguments_on_the_host():
    """cstr_rollout_step_f32, cstr_replay_gather_packed_f32, cstr_linear_act_fwd_gather_f32: argument checks fail before anything is
    dereferenced or launched."""
    lib = nv.lib()
    null, fake = C.c_void_p(None), C.c_void_p(0x1000)
    i64, u64 = C.c_int64, C.c_uint64
    coef = nv.default_coef()
    ring = nv.Ring(0x1000, 0x1000, 0x1000, 0x1000, 0x1000, 0x1000, 4, 64, 4, 2)
    lo, hi = (C.c_float * 2)(-1, -1), (C.c_float * 2)(1, 1)
    net = nv.PolicyMlp(4, 64, 64, 2, 1, 0, 0, 1, 0x1000, 0x1000, 0x1000, 0x1000, 0x1000, 0x1000, 0x1000)

    def rollout(net_, rng, pcg, mt, batch, idx, ring_=ring, ring_ctl=fake):
        return lib.cstr_rollout_step_f32(C.byref(net_), fake, i64(4), rng, C.byref(coef), 0, C.byref(ring_), ring_ctl, fake, fake, 1, lo, hi, null,
                                         null, pcg, null, null, null, null, null, null, mt, i64(batch), idx, null)

    assert rollout(net, fake, null, null, 0, null) == -1            # no reset source
    assert rollout(net, null, fake, null, 0, null) == -1            # sampling head without its Philox stream
    assert rollout(net, fake, fake, fake, 256, null) == -1          # index draw without its output buffer
    assert rollout(net, fake, fake, fake, 0, fake) == -1            # ... without a batch size
    assert rollout(net, fake, fake, null, 0, null, ring_ctl=null) == -1
    no_copy = nv.PolicyMlp(4, 64, 64, 2, 1, 0, 0, 1, 0x1000, 0x1000, 0x1000, 0x1000, 0x1000, 0x1000, None)
    assert rollout(no_copy, fake, fake, null, 0, null) == -2        # needs the tile-major W2 copy
    wide = nv.PolicyMlp(32, 64, 64, 2, 1, 0, 0, 1, 0x1000, 0x1000, 0x1000, 0x1000, 0x1000, 0x1000, 0x1000)
    assert lib.cstr_rollout_step_f32(C.byref(wide), fake, i64(32), fake, C.byref(coef), 0, C.byref(ring), fake, fake, fake, 1, lo, hi, null, null,
                                     fake, null, null, null, null, null, null, null, i64(0), null, null) == -2  # first layer wider than one k chunk
    four = nv.PolicyMlp(4, 64, 64, 4, 1, 1, 2, 1, 0x1000, 0x1000, 0x1000, 0x1000, 0x1000, 0x1000, 0x1000)
    assert rollout(four, null, fake, null, 0, null) == -1           # policy's action width != the ring's
    # gather launch
    assert lib.cstr_replay_gather_packed_f32(C.byref(ring), fake, 1, null, u64(0), null, i64(8), fake, fake, null, fake, fake, null, null, null) == -1
    assert lib.cstr_replay_gather_packed_f32(C.byref(ring), null, 1, null, u64(0), fake, i64(8), fake, fake, null, fake, fake, null, null, null) == -1
    assert lib.cstr_replay_gather_packed_f32(C.byref(ring), fake, 1, null, u64(0), fake, i64(1 << 20), fake, fake, null, fake, fake, null, null, null) == -2
    odd = nv.Ring(0x1000, 0x1000, 0x1000, 0x1000, 0x1000, 0x1000, 4, 64, 5, 2)
    assert lib.cstr_replay_gather_packed_f32(C.byref(odd), fake, 1, null, u64(0), fake, i64(8), fake, fake, null, fake, fake, null, null, null) == -2
    # first layer with the gather inside
    def gl(idx, w, act, ring_ctl=fake, batch=8, ring_=ring):
        return lib.cstr_linear_act_fwd_gather_f32(C.byref(ring_), ring_ctl, 1, null, u64(0), idx, i64(batch), 1, w, fake, act, fake, i64(64), fake, fake,
                                                  null, fake, fake, null)

    assert gl(null, fake, 1) == -1 and gl(fake, null, 1) == -1 and gl(fake, fake, 7) == -1 and gl(fake, fake, 1, ring_ctl=null) == -1
    assert gl(fake, C.c_void_p(0x1004), 1) == -1                   # weight rows are read as 16-byte vectors
    assert gl(fake, fake, 1, batch=1 << 20) == -2 and gl(fake, fake, 1, ring_=odd) == -2


def test_chain_entry_points_reject_bad_arguments_on_the_host():
    """The row-chain ABI (include/cstr_rl_hip.h, "row-chain kernels") validates shapes, layouts and pointers before any launch."""
    lib = nv.lib()
    null, fake = C.c_void_p(None), C.c_void_p(0x1000)
    actor = nv.SacActorNet(4, 2, 256, 256, 0x1000, 0x1000, 0x1000, 0x1000, 0x1000, 0x1000)
    args = lambda a, batch, tiles, mode=0, hn=4: (C.byref(a), null, null, 0, null, C.c_int64(batch), null, fake, fake, null, null, fake, fake, fake,  # noqa: E731
                                                     null, C.c_uint64(0), null, mode, hn, tiles, null)
    assert lib.cstr_sac_actor_chain_fwd_f32(*args(nv.SacActorNet(5, 2, 256, 256, 0x1000, 0x1000, 0x1000, 0x1000, 0x1000, 0x1000), 256, 2)) == -2  # layout
    assert lib.cstr_sac_actor_chain_fwd_f32(*args(actor, 250, 2)) == -2        # batch not a multiple of 16
    assert lib.cstr_sac_actor_chain_fwd_f32(*args(actor, 256, 3)) == -2        # tiles
    assert lib.cstr_sac_actor_chain_fwd_f32(*args(actor, 256, 2, mode=7)) == -1
    assert lib.cstr_sac_actor_chain_fwd_f32(*args(actor, 256, 2, hn=3)) == -1  # head outputs: A or 2A
    wide = nv.SacActorNet(4, 2, 640, 256, 0x1000, 0x1000, 0x1000, 0x1000, 0x1000, 0x1000)
    assert lib.cstr_sac_actor_chain_fwd_f32(*args(wide, 256, 2)) == -2         # wider than CSTR_CHAIN_MAX_WIDTH
    net = nv.ChainNet(*([0x1000] * 10), 0, 0)
    nets = (nv.ChainNet * 1)(net)
    assert lib.cstr_q_chain_fwd_f32(nets, 17, 6, 4, 256, 256, C.c_int64(256), null, 2, null) == -1  # more than 16 networks
    assert lib.cstr_q_chain_fwd_f32(nets, 1, 6, 3, 256, 256, C.c_int64(256), null, 2, null) == -2   # (3, 3) is not a layout
    bad_role = (nv.ChainNet * 1)(nv.ChainNet(*([0x1000] * 10), 2, 0))
    assert lib.cstr_q_chain_fwd_f32(bad_role, 1, 6, 4, 256, 256, C.c_int64(256), null, 2, null) == -1  # a finalising role without `fin`
    root = nv.ChainRoot()
    root.mode, root.batch, root.n_parts = 9, 256, 4
    assert lib.cstr_q_chain_bwd_f32(nets, 1, C.byref(root), 6, 4, 256, 256, null, null, null, 2, null) == -1  # mode
    root.mode = 1
    assert lib.cstr_q_chain_bwd_f32(nets, 1, C.byref(root), 6, 4, 256, 256, null, null, null, 2, null) == -1  # TD root needs two networks
    assert lib.cstr_sac_actor_chain_bwd_f32(C.byref(actor), null, 2, 8, fake, fake, fake, fake, fake, fake, fake, fake, fake, C.c_int64(256), 0, 1, null) == -1
    assert lib.cstr_linear_bwd_weight_adam_sets_f32(null, 1, null, 1, null, 0, null) == -1
