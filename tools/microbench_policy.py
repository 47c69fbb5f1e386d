#!/usr/bin/env python3
"""Launch the rollout's whole-policy kernel alone at the bench shape (4096 rows, 4 -> 256 -> 256 -> 2x2, tile-major copy of the
hidden layer like the product path), for rocprofv3 --pmc / --kernel-trace passes (tools/pmc_policy.sh)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "pytorch-rl-enhancedstablebaselines_amd")):
    sys.path.insert(0, p)
import torch as th  # noqa: E402

from core.common import hip_ops  # noqa: E402

if __name__ == "__main__":
    m, k0, h, a = 4096, 4, 256, 2
    r = lambda *s: th.randn(*s, device="cuda")  # noqa: E731
    x, w1, b1, w2, b2, w3, b3 = r(m, k0), r(h, k0), r(h), r(h, h) / 16, r(h), r(2 * a, h) / 16, r(2 * a)
    ctl, act, tiles = hip_ops.new_rng_ctl(1, "cuda"), th.empty(m, a, device="cuda"), hip_ops.policy_swizzle(w2)
    for _ in range(200):
        hip_ops.policy_rows_fwd(x, w1, b1, w2, b2, w3, b3, 1, 0, 0, act, rng_ctl=ctl, w2_swz=tiles, defer_rng_advance=True)
    th.cuda.synchronize()
