#!/usr/bin/env python3
"""Per-phase timeline of the policy kernel from in-kernel s_memtime stamps (diagnostic build: `make -C .../csrc diag`,
run with CSTR_LIB_PATH=tools/ab/libcstr_rl_hip_diag.so). Stamps per wave: 0 start | 1 layer 1 (or noise) done | 2 past barrier |
3 layer 2 done | 4 past barrier | 5 head partials combined (past barrier) | 6 tail done | 7 ticket done. Prints, in shader cycles
relative to the earliest stamp 0 of the launch: median / max over workgroups of each wave's stamps."""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("CSTR_LIB_PATH", os.path.join(ROOT, "tools", "ab", "libcstr_rl_hip_diag.so"))
for p in (ROOT, os.path.join(ROOT, "pytorch-rl-enhancedstablebaselines_amd")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import torch as th  # noqa: E402

from core import _native as nv  # noqa: E402
from core.common import hip_ops  # noqa: E402

if __name__ == "__main__":
    defer = "--defer" in sys.argv
    shape = (4096, 4, 400, 300, 2, 1) if "--td3" in sys.argv else (4096, 4, 256, 256, 2, 0)
    m, k0, h1, h2, a, head = shape
    r = lambda *s: th.randn(*s, device="cuda")  # noqa: E731
    x, w1, b1, w2, b2 = r(m, k0), r(h1, k0), r(h1), r(h2, h1) / h1 ** 0.5, r(h2)
    n_out = 2 * a if head == 0 else a
    w3, b3 = r(n_out, h2) / 16, r(n_out)
    ctl, act, tiles = hip_ops.new_rng_ctl(1, "cuda"), th.empty(m, a, device="cuda"), hip_ops.policy_swizzle(w2)
    rollout = "--rollout" in sys.argv  # the one-launch rollout (policy + collect step + index draw): stamp 7 of waves 0-1 = tail done
    if rollout:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from test_rollout_step import _World

        w = _World(hip_ops, m, 4, 244, 256, h1, h2, head, seed=1, max_steps=400)
        w.step_count.zero_()
    for _ in range(20):
        if rollout:
            w.step_fused(None)
        else:
            hip_ops.policy_rows_fwd(x, w1, b1, w2, b2, w3, b3, 1, head, 2 if head else 0, act, rng_ctl=ctl if head == 0 else None, w2_swz=tiles,
                                    defer_rng_advance=defer)
    th.cuda.synchronize()
    n_blocks = m // 16
    words = n_blocks * 16 * 8  # device layout: [(block * 16 + wave) * 8 + stamp]
    buf = (C.c_uint64 * words)()
    lib = nv.lib()
    lib.cstr_diag_policy_stamps.argtypes = [C.c_void_p, C.c_int64]
    assert lib.cstr_diag_policy_stamps(buf, words) == 0
    st = np.frombuffer(buf, dtype=np.uint64).reshape(n_blocks, 16, 8)[:, :8, :].astype(np.int64)
    rel = st - st[:, :, 0].min(axis=1)[:, None, None]  # per workgroup (the counter differs between XCDs)
    out = dict(shape=shape, defer=defer, rollout=rollout, clock_note="shader cycles (s_memtime); 2.4 GHz nominal -> 2400 cycles = 1 us")
    names = ["start", "l1_done", "bar1", "l2_done", "bar2", "head_done", "end", "noise_done(wave7)/tail_done(waves0-1,rollout)"]
    for i, nm in enumerate(names):
        v = rel[:, :, i]
        out[nm] = dict(per_wave_median=[int(np.median(v[:, w])) for w in range(8)])
    # second bank (h <= 256 instantiation): inside layer 2 -- 0 at the loop's entry, 1-4 behind the MFMAs of each k stage, 5 epilogue done
    st2 = np.frombuffer(buf, dtype=np.uint64).reshape(n_blocks, 16, 8)[:, 8:, :].astype(np.int64)
    rel2 = st2 - st[:, :, 0].min(axis=1)[:, None, None]
    for i, nm in enumerate(["l2_entry", "l2_stage0", "l2_stage1", "l2_stage2", "l2_stage3", "l2_epilogue"]):
        out[nm] = dict(per_wave_median=[int(np.median(rel2[:, w, i])) for w in range(8)])
    print(json.dumps(out))
