#!/usr/bin/env python3
"""What does a by-value struct argument (a pointer table indexed by blockIdx.z) cost a small launch against scalar arguments that
arrive preloaded in SGPRs? The same Linear + ReLU at 256 x 256 x 256 and at the critics' first-layer shape, graph-replayed chains."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "pytorch-rl-enhancedstablebaselines_amd"), os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)
import torch as th  # noqa: E402

from core.common import hip_ops as ops  # noqa: E402
from rollout_ab import timed  # noqa: E402

if __name__ == "__main__":
    for m, n, k in ((256, 256, 256), (256, 256, 8), (512, 256, 4)):
        x, w, b = th.randn(m, k, device="cuda"), th.randn(4, n, k, device="cuda") / 16, th.zeros(4, n, device="cuda")
        y = th.empty(4, m, n, device="cuda")
        print(f"{m} x {n} x {k}: scalar arguments, 1 group {timed(lambda: ops.linear_act_fwd(x, w[0], b[0], 1, out=y[0])):6.2f} us | "
              f"pointer table, 1 set {timed(lambda: ops.linear_act_fwd_sets([(x, w[0], b[0], y[0])], 1)):6.2f} us | "
              f"scalar, 4 groups {timed(lambda: ops.linear_act_fwd(x.unsqueeze(0).expand(4, -1, -1), w, b, 1, out=y)):6.2f} us | "
              f"pointer table, 4 sets {timed(lambda: ops.linear_act_fwd_sets([(x, w[g], b[g], y[g]) for g in range(4)], 1)):6.2f} us")
