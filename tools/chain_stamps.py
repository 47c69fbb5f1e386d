#!/usr/bin/env python3
"""Per-phase timeline of the chain kernels from in-kernel s_memtime stamps (diagnostic build: `make -C .../csrc diag`, run with
CSTR_LIB_PATH=tools/ab/libcstr_rl_hip_diag.so). Stamps per wave: 0 entry | 1 prologue done (before the first barrier) | 2 past it |
3 panel ready (layer 1 / dz2 recomputed, past the barrier) | 4 MFMAs done | 5 split-K combined | 6 partials reduced (past the last
barrier) | 7 stores issued. Prints the median over workgroups (and the slowest workgroup) of each phase end, in ns after the launch's
earliest stamp of the same XCD-agnostic per-workgroup origin (stamp 0 of the workgroup's first wave)."""
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
from core.common.vec_env import CSTRVecEnv  # noqa: E402
from core.sac import SAC  # noqa: E402
from tools.chain_probe import chain_launches  # noqa: E402

KERNEL_OF = dict(sac_actor_chain_fwd=0, q_chain_fwd_4nets=1, q_chain_bwd_td=2, q_chain_fwd_2nets=1, q_chain_bwd_actor=2, sac_actor_chain_bwd=3)
NAMES = ["entry", "prologue", "barrier1", "panel", "mfma", "combine", "partials", "stores"]

if __name__ == "__main__":
    B = 256
    arch = [int(v) for v in os.environ.get("ARCH", "256,256").split(",")]
    model = SAC("MlpPolicy", CSTRVecEnv(4096), seed=0, batch_size=B, policy_kwargs=dict(net_arch=arch))
    model.learn(4096 * 4)
    fns = chain_launches(model, B)
    lib = nv.lib()
    lib.cstr_diag_chain_stamps.argtypes = [C.c_void_p, C.c_int64]
    words = 4 * 1024 * 4 * 8
    buf = (C.c_uint64 * words)()
    out = {}
    ghz = 0.1  # s_memtime on gfx950 ticks at 100 MHz
    for name, (fn, _) in fns.items():
        for _ in range(5):
            fn()
        # a different kernel in between, like in the iteration's graph (cold operands)
        th.cuda.synchronize()
        fn()
        th.cuda.synchronize()
        assert lib.cstr_diag_chain_stamps(buf, words) == 0
        st = np.frombuffer(buf, dtype=np.uint64).reshape(4, 1024, 4, 8)[KERNEL_OF[name]].astype(np.int64)
        live = st[:, 0, 0] > 0
        st = st[live]
        t0 = st[:, :, 0].min(axis=1)  # the workgroup's own entry
        launch0 = st[:, :, 0].min()
        rel = (st - t0[:, None, None]) / ghz
        row = {}
        for i, nm in enumerate(NAMES):
            v = rel[:, :, i].max(axis=1)  # the workgroup's last wave
            v = v[(st[:, :, i].max(axis=1) > 0)]
            if v.size:
                row[nm] = dict(median_ns=int(np.median(v)), max_ns=int(v.max()))
        row["entry_spread_ns"] = int((t0.max() - launch0) / ghz)
        row["workgroups"] = int(live.sum())
        out[name] = row
        print(name, {k: (int(v["median_ns"] / 10) if isinstance(v, dict) else v) for k, v in row.items() if k != "entry_spread_ns"})
    tag = "" if arch == [256, 256] else "_" + "x".join(str(v) for v in arch)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", f"r03_chain_phase_stamps{tag}.json"), "w"), indent=1)
