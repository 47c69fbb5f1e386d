#!/usr/bin/env python3
"""Phase stamps of the dW / db + Adam launch (diagnostic build, `make -C .../csrc diag`): a short SAC run under hipGraph replay, then the
stamps the LAST launches left (blocks < 300: the actor's launch, the others: the critic's). Per workgroup, cycles after its first
stamp: 1 operands and moments requested | 2 rows loaded + MFMAs done (per wave) | 3 past the split-M barrier | 4 Adam scalars read
(first wave) | 5 stores issued."""
import ctypes as C
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

if __name__ == "__main__":
    n = 4096
    model = SAC("MlpPolicy", CSTRVecEnv(n), seed=0, learning_starts=n * 2)
    model.enable_graph_capture(True)
    model.learn(n * 40)
    th.cuda.synchronize()
    words = 4096 * 16 * 8
    buf = (C.c_uint64 * words)()
    lib = nv.lib()
    lib.cstr_diag_policy_stamps.argtypes = [C.c_void_p, C.c_int64]
    assert lib.cstr_diag_policy_stamps(buf, words) == 0
    st = np.frombuffer(buf, dtype=np.uint64).reshape(4096, 16, 8)[:, :4, :6].astype(np.int64)
    for name, sl in (("actor launch (blocks 0-255)", slice(0, 256)), ("critic launch (blocks 320-575)", slice(320, 576))):
        s = st[sl]
        base = s[:, :, 0].min(axis=1)[:, None, None]
        rel = s - base
        launch0 = s[:, :, 0].min()
        print(name)
        for i, nm in enumerate(["entry", "requested", "mfma_done", "past_barrier", "adam_scalars", "stores"]):
            v = rel[:, :, i]
            ok = s[:, :, i] > 0
            print(f"  {nm:14s} per wave median", [int(np.median(v[:, w][ok[:, w]])) if ok[:, w].any() else -1 for w in range(4)])
