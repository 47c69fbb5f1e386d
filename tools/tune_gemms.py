#!/usr/bin/env python3
"""Re-tune a PyTorch TunableOp results file with GPU-side timing (MI355X).

PyTorch's TunableOp times every candidate with eager launches; for this path's GEMMs (batch 256, width 256..400) every
good rocBLAS solution finishes faster than the host can launch (~6.4 us), so its choice among them is noise. The
training loop replays the iteration from a hipGraph, where the GPU-side duration is what counts. This tool takes the
problem keys of an existing results file (written by `bench.py --tunable 1`, which records every GEMM shape the learners
issue), enumerates rocBLAS' solutions for each problem through the same librocblas PyTorch uses
(rocblas_gemm_ex_get_solutions / rocblas_gemm_strided_batched_ex_get_solutions), times each one as the issue interval of
graph-replayed launches, checks its result against the default solution, and writes the winners in TunableOp's format.
`core/common/blas.py` loads the committed result (core/common/tunableop_gfx950.csv) with tuning disabled.

usage: python tools/tune_gemms.py in.csv out.csv [--reps 40]
"""
import argparse
import ctypes as C
import os
import re
import sys
import time

import torch as th

OP_N, OP_T, F32, ALGO_STD, ALGO_IDX = 111, 112, 151, 0, 1

KEY = re.compile(r"^(?P<ta>[tn])(?P<tb>[tn])_(?P<m>\d+)_(?P<n>\d+)_(?P<k>\d+)(?:_B_(?P<b>\d+))?_ld_(?P<lda>\d+)_(?P<ldb>\d+)_(?P<ldc>\d+)$")


def load_rocblas():
    path = os.path.join(os.path.dirname(th.__file__), "lib", "librocblas.so")
    lib = C.CDLL(path if os.path.exists(path) else "librocblas.so")
    h = C.c_void_p()
    assert lib.rocblas_create_handle(C.byref(h)) == 0
    return lib, h


class Problem:
    def __init__(self, key):
        m = KEY.match(key)
        if not m:
            raise ValueError(key)
        g = m.groupdict()
        self.key = key
        self.ta, self.tb = g["ta"], g["tb"]
        self.m, self.n, self.k = int(g["m"]), int(g["n"]), int(g["k"])
        self.lda, self.ldb, self.ldc = int(g["lda"]), int(g["ldb"]), int(g["ldc"])
        self.batch = int(g["b"]) if g["b"] else 0
        # column-major operands: A is lda x (k if 'n' else m), B is ldb x (n if 'n' else k), C is ldc x n
        self.sa = self.lda * (self.k if self.ta == "n" else self.m)
        self.sb = self.ldb * (self.n if self.tb == "n" else self.k)
        self.sc = self.ldc * self.n
        nb = max(self.batch, 1)
        gen = th.Generator(device="cuda").manual_seed(1)
        self.a = th.randn(nb * self.sa, device="cuda", generator=gen)
        self.b = th.randn(nb * self.sb, device="cuda", generator=gen)
        self.c = th.zeros(nb * self.sc, device="cuda")
        self.alpha, self.beta = C.c_float(1.0), C.c_float(0.0)

    def _common(self):
        p = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
        op = lambda c: OP_T if c == "t" else OP_N  # noqa: E731
        head = [op(self.ta), op(self.tb), self.m, self.n, self.k, C.byref(self.alpha)]
        if self.batch:
            s = C.c_int64
            return head + [p(self.a), F32, self.lda, s(self.sa), p(self.b), F32, self.ldb, s(self.sb), C.byref(self.beta),
                           p(self.c), F32, self.ldc, s(self.sc), p(self.c), F32, self.ldc, s(self.sc), self.batch, F32]
        return head + [p(self.a), F32, self.lda, p(self.b), F32, self.ldb, C.byref(self.beta), p(self.c), F32, self.ldc,
                       p(self.c), F32, self.ldc, F32]

    def solutions(self, lib, h):
        fn = lib.rocblas_gemm_strided_batched_ex_get_solutions if self.batch else lib.rocblas_gemm_ex_get_solutions
        n = C.c_int(0)
        if fn(h, *self._common(), ALGO_IDX, C.c_uint32(0), None, C.byref(n)) != 0 or n.value <= 0:
            return []
        arr = (C.c_int * n.value)()
        if fn(h, *self._common(), ALGO_IDX, C.c_uint32(0), arr, C.byref(n)) != 0:
            return []
        return list(arr[:n.value])

    def run(self, lib, h, sol):
        fn = lib.rocblas_gemm_strided_batched_ex if self.batch else lib.rocblas_gemm_ex
        algo, idx = (ALGO_STD, 0) if sol is None else (ALGO_IDX, sol)
        return fn(h, *self._common(), algo, C.c_int32(idx), C.c_uint32(0))


def graph_time_us(fn, reps):
    side = th.cuda.Stream()
    side.wait_stream(th.cuda.current_stream())
    g = th.cuda.CUDAGraph()
    with th.cuda.stream(side):
        g.capture_begin(capture_error_mode="thread_local")
        for _ in range(reps):
            fn()
        g.capture_end()
    th.cuda.current_stream().wait_stream(side)
    g.replay()
    th.cuda.synchronize()
    best = float("inf")
    for _ in range(3):
        e0, e1 = th.cuda.Event(enable_timing=True), th.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        e1.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / reps)
    return best


def tune(lib, h, prob, reps):
    stream = th.cuda.current_stream()

    def call(sol):
        # the handle follows whichever stream is current (the capture stream inside graph_time_us)
        lib.rocblas_set_stream(h, C.c_void_p(th.cuda.current_stream().cuda_stream))
        return prob.run(lib, h, sol)

    if call(None) != 0:
        # e.g. n == 1 keys carry a degenerate leading dimension that PyTorch repairs after building the key: leave as is
        lib.rocblas_set_stream(h, C.c_void_p(stream.cuda_stream))
        return None
    th.cuda.synchronize()
    ref = prob.c.clone()
    scale = float(ref.abs().max()) or 1.0
    results = [("Default", graph_time_us(lambda: call(None), reps))]
    for sol in prob.solutions(lib, h):
        prob.c.zero_()
        if call(sol) != 0:
            continue
        th.cuda.synchronize()
        if not th.isfinite(prob.c).all() or float((prob.c - ref).abs().max()) > 1e-4 * scale:
            continue
        try:
            results.append((f"Gemm_Rocblas_{sol}", graph_time_us(lambda s=sol: call(s), reps)))
        except RuntimeError:
            th.cuda.synchronize()
            continue
    lib.rocblas_set_stream(h, C.c_void_p(stream.cuda_stream))
    results.sort(key=lambda r: r[1])
    return results


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("src")
    ap.add_argument("dst")
    ap.add_argument("--reps", type=int, default=40)
    args = ap.parse_args()
    lib, h = load_rocblas()
    header, rows = [], []
    for line in open(args.src):
        line = line.strip()
        if not line:
            continue
        (header if line.startswith("Validator") else rows).append(line)
    out, t0 = list(header), time.time()
    for i, line in enumerate(rows):
        op, key, old_name, old_ms = line.split(",")
        if "float" not in op or not KEY.match(key):
            out.append(line)
            continue
        prob = Problem(key)
        res = tune(lib, h, prob, args.reps)
        if res is None:
            print(f"[{i + 1}/{len(rows)}] {op} {key}: not callable with the key's leading dimensions, kept {old_name}", flush=True)
            out.append(line)
            continue
        by_name = dict(res)
        best_name, best_us = res[0]
        print(f"[{i + 1}/{len(rows)}] {op} {key}: {len(res)} candidates, best {best_name} {best_us:.2f} us; default "
              f"{by_name['Default']:.2f} us; eager-tuned {old_name} {by_name.get(old_name, float('nan')):.2f} us "
              f"({time.time() - t0:.0f} s)", flush=True)
        out.append(f"{op},{key},{best_name},{best_us / 1e3:.6g}")
    with open(args.dst, "w") as f:
        f.write("\n".join(out) + "\n")
    print("wrote", args.dst, file=sys.stderr)


if __name__ == "__main__":
    main()
