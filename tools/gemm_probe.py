#!/usr/bin/env python3
"""GPU probe: time the MLP GEMM shapes of the SAC step under the BLAS back ends PyTorch-ROCm offers
(hipBLASLt default, rocBLAS, TunableOp). Decides which one bench.py / the learner should prefer."""
import json
import os
import sys
import time

import torch as th

BMM = [(2, 256, 256, 256), (2, 256, 256, 6), (2, 256, 1, 256)]
SHAPES = [  # (M, N, K) of y[M,N] = x[M,K] @ W[N,K]^T (+ bias) and the two backward GEMMs of each
    (4096, 256, 4), (4096, 256, 256), (4096, 2, 256),
    (256, 256, 4), (256, 256, 6), (256, 256, 256), (256, 1, 256), (256, 2, 256),
]


def t_us(fn, n=200):
    """issue interval of graph-replayed launches (GPU-side cost; the eager host gap would hide it)"""
    for _ in range(10):
        fn()
    th.cuda.synchronize()
    side = th.cuda.Stream()
    side.wait_stream(th.cuda.current_stream())
    g = th.cuda.CUDAGraph()
    with th.cuda.stream(side):
        g.capture_begin()
        for _ in range(n):
            fn()
        g.capture_end()
    th.cuda.current_stream().wait_stream(side)
    g.replay()
    th.cuda.synchronize()
    e0, e1 = th.cuda.Event(enable_timing=True), th.cuda.Event(enable_timing=True)
    e0.record()
    g.replay()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


def probe(tag):
    out = {}
    for (m, n, k) in SHAPES:
        x = th.randn(m, k, device="cuda")
        w = th.randn(n, k, device="cuda")
        b = th.randn(n, device="cuda")
        g = th.randn(m, n, device="cuda")
        fwd = t_us(lambda: th.mm(x, w.t()))
        dx = t_us(lambda: g @ w)          # grad_input  [M,K] = g[M,N] @ W[N,K]
        dw = t_us(lambda: g.t() @ x)      # grad_weight [N,K] = g^T[N,M] @ x[M,K]
        out[f"{m}x{n}x{k}"] = dict(fwd=round(fwd, 2), dx=round(dx, 2), dw=round(dw, 2))
    for (g_, m, n, k) in BMM:
        x = th.randn(g_, m, k, device="cuda")
        w = th.randn(g_, n, k, device="cuda")
        out[f"bmm{g_}x{m}x{n}x{k}"] = dict(fwd=round(t_us(lambda: th.bmm(x, w.transpose(1, 2))), 2))
    print(tag, json.dumps(out), flush=True)
    return out


if __name__ == "__main__":
    mode = sys.argv[1] if len(sys.argv) > 1 else "default"
    if mode == "rocblas":
        th.backends.cuda.preferred_blas_library("cublas")
    elif mode == "ck":
        th.backends.cuda.preferred_blas_library("ck")
    elif mode == "hipblaslt":
        th.backends.cuda.preferred_blas_library("cublaslt")
    elif mode == "tunable":
        th.cuda.tunable.enable(True)
        th.cuda.tunable.tuning_enable(True)
        th.cuda.tunable.set_filename(os.path.join("gpurun_out", "tunableop_probe.csv"))
    print("preferred:", th.backends.cuda.preferred_blas_library())
    t0 = time.time()
    probe(mode)
    if mode == "tunable":
        probe(mode + "_2nd")
    print("elapsed", round(time.time() - t0, 1))
