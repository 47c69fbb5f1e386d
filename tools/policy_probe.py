#!/usr/bin/env python3
"""GPU probe: the whole policy network in one launch (cstr_policy_rows_fwd_f32) vs the layer-by-layer inference path, timed
as graph-replayed launches."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "pytorch-rl-enhancedstablebaselines_amd")):
    sys.path.insert(0, p)
import torch as th  # noqa: E402

from core.common import blas, fused, hip_ops  # noqa: E402
from tools.gemm_probe import t_us  # noqa: E402

if __name__ == "__main__":
    blas.configure()
    for (m, k0, h1, h2, a) in [(4096, 4, 256, 256, 2), (256, 4, 256, 256, 2), (1024, 4, 256, 256, 2), (4096, 8, 400, 300, 2)]:
        r = lambda *s: th.randn(*s, device="cuda")  # noqa: E731
        x, w1, b1, w2, b2, w3, b3 = r(m, k0), r(h1, k0), r(h1), r(h2, h1) / 16, r(h2), r(2 * a, h2) / 16, r(2 * a)
        ctl = hip_ops.new_rng_ctl(1, "cuda")
        act, lp, eps, params = th.empty(m, a, device="cuda"), th.empty(m, device="cuda"), th.empty(m, a, device="cuda"), th.empty(m, 2 * a, device="cuda")

        def layers():
            h = fused._linear_fwd(x, w1, b1, 1)
            h = fused._linear_fwd(h, w2, b2, 1)
            if m <= 1024:
                hip_ops.gaussian_head_gemm_fwd(h, w3, b3, params, eps, ctl, act, lp)
            else:
                hip_ops.gaussian_head_fwd_(th.mm(h, w3.t()), b3, eps, ctl, act, lp)

        one = t_us(lambda: hip_ops.policy_rows_fwd(x, w1, b1, w2, b2, w3, b3, 1, 0, 0, act, rng_ctl=ctl, logp=lp))
        print(f"M={m} {k0}->{h1}->{h2}->2x{a}: layer by layer {t_us(layers):6.2f} us   one launch {one:6.2f} us", flush=True)
