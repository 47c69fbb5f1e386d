#!/usr/bin/env python3
"""GPU probe: Linear + bias + activation forward as (rocBLAS GEMM from the tuned table + epilogue launch) vs the one-launch
f32-MFMA kernel cstr_linear_act_fwd_f32, both timed as graph-replayed launches (the way the training loop issues them)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "pytorch-rl-enhancedstablebaselines_amd")):
    sys.path.insert(0, p)
import torch as th  # noqa: E402

from core.common import blas, hip_ops  # noqa: E402
from tools.gemm_probe import t_us  # noqa: E402

SHAPES = [(0, 256, 256, 4), (0, 256, 256, 256), (0, 4096, 256, 4), (0, 4096, 256, 256), (2, 256, 256, 6), (2, 256, 256, 256),
          (0, 256, 400, 4), (0, 256, 300, 400), (2, 256, 400, 6), (2, 256, 300, 400), (0, 4096, 400, 4), (0, 4096, 300, 400)]

if __name__ == "__main__":
    blas.configure()
    for (g, m, n, k) in SHAPES:
        gg = max(g, 1)
        x = th.randn(gg, m, k, device="cuda")
        w, b = th.randn(gg, n, k, device="cuda"), th.randn(gg, n, device="cuda")
        if g == 0:
            x, w, b = x[0], w[0], b[0]
            two = t_us(lambda: hip_ops.bias_act_fwd_(th.mm(x, w.t()), b, 1))
        else:
            two = t_us(lambda: hip_ops.bias_act_fwd_(th.bmm(x, w.transpose(1, 2)), b, 1))
        one = t_us(lambda: hip_ops.linear_act_fwd(x, w, b, 1))
        print(f"G={g} M={m} N={n} K={k}: GEMM + epilogue {two:6.2f} us   fused MFMA kernel {one:6.2f} us", flush=True)
    print("input gradient + activation / bias gradient of the layer below:")
    for (g, m, n, k) in [(0, 256, 256, 256), (2, 256, 256, 256), (0, 256, 4, 256), (0, 256, 300, 400), (2, 256, 300, 400), (0, 256, 2, 300)]:
        gg = max(g, 1)
        gz, w = th.randn(gg, m, n, device="cuda"), th.randn(gg, n, k, device="cuda")
        y, gb = th.relu(th.randn(gg, m, k, device="cuda")), th.empty(gg, k, device="cuda")
        if g == 0:
            gz, w, y, gb = gz[0], w[0], y[0], gb[0]

        def two(bias_grad):
            dx = th.bmm(gz, w) if g else th.mm(gz, w)
            out = th.empty_like(dx)
            hip_ops.bias_act_bwd(dx, y, 1, out, gb if bias_grad else None)

        def fused_plus_colsum():
            dz = hip_ops.linear_bwd_input(gz, w, y, 1)
            hip_ops.bias_act_bwd(dz, None, 0, dz, gb)

        print(f"G={g} M={m} N={n} K={k}: GEMM + bias_act_bwd {t_us(lambda: two(True)):6.2f} us | fused + column sums "
              f"{t_us(fused_plus_colsum):6.2f} us || without bias gradient: {t_us(lambda: two(False)):6.2f} us | fused "
              f"{t_us(lambda: hip_ops.linear_bwd_input(gz, w, y, 1)):6.2f} us", flush=True)
    print("weight + bias gradient:")
    for (g, m, n, k) in [(0, 256, 256, 256), (0, 256, 256, 4), (2, 256, 256, 6), (2, 256, 256, 256), (0, 256, 300, 400), (0, 256, 400, 4)]:
        gg = max(g, 1)
        dzz, xx = th.randn(gg, m, n, device="cuda"), th.randn(gg, m, k, device="cuda")
        dw, db = th.empty(gg, n, k, device="cuda"), th.empty(gg, n, device="cuda")
        if g == 0:
            dzz, xx, dw, db = dzz[0], xx[0], dw[0], db[0]

        def two():
            th.bmm(dzz.transpose(1, 2), xx, out=dw) if g else th.mm(dzz.t(), xx, out=dw)
            hip_ops.bias_act_bwd(dzz, None, 0, dzz, db)

        print(f"G={g} M={m} N={n} K={k}: GEMM + column sums {t_us(two):6.2f} us   fused MFMA kernel "
              f"{t_us(lambda: hip_ops.linear_bwd_weight(dzz, xx, dw, db)):6.2f} us   (GEMM alone "
              f"{t_us(lambda: th.bmm(dzz.transpose(1, 2), xx, out=dw) if g else th.mm(dzz.t(), xx, out=dw)):6.2f} us)", flush=True)
    print("SAC actor head (Linear 256 -> 4, sampling):")
    for b in (256, 4096):
        h, w, bias = th.randn(b, 256, device="cuda"), th.randn(4, 256, device="cuda") / 16, th.zeros(4, device="cuda")
        p, e, a, lp = th.empty(b, 4, device="cuda"), th.empty(b, 2, device="cuda"), th.empty(b, 2, device="cuda"), th.empty(b, device="cuda")
        ctl = hip_ops.new_rng_ctl(1, "cuda")
        two = t_us(lambda: hip_ops.gaussian_head_fwd_(th.mm(h, w.t()), bias, e, ctl, a, lp))
        one = t_us(lambda: hip_ops.gaussian_head_gemm_fwd(h, w, bias, p, e, ctl, a, lp))
        print(f"B={b}: GEMM + head kernel {two:6.2f} us   head kernel with the Linear inside {one:6.2f} us", flush=True)
