#!/bin/bash
# Re-create core/common/tunableop_gfx950.csv on an MI355X: (1) record every GEMM shape the learners issue (bench.py --tunable 1
# lets PyTorch's TunableOp log them), (2) pick the fastest rocBLAS solution per shape by graph-replayed timing
# (tools/tune_gemms.py). Run from the repo root through gpurun; the result lands in gpurun_out/tuned.csv.
set -e
mkdir -p gpurun_out
rm -f gpurun_out/tunableop_results*.csv
B="python bench.py --tunable 1 --no-cpu-baseline --no-roofline --no-variant --steps 20 --warmup 10"
export CSTR_TUNABLEOP_FILE=0
$B > gpurun_out/k1.log 2>&1
$B --algo td3 > gpurun_out/k2.log 2>&1
$B --algo maddpg --n-envs 1024 > gpurun_out/k3.log 2>&1
$B --obs-dim 8 > gpurun_out/k4.log 2>&1
$B --algo td3 --obs-dim 8 > gpurun_out/k5.log 2>&1
cp gpurun_out/tunableop_results.csv gpurun_out/keys.csv
timeout -k 10 900 python tools/tune_gemms.py gpurun_out/keys.csv gpurun_out/tuned.csv > gpurun_out/tune.log 2>&1
tail -2 gpurun_out/tune.log | cut -c1-200
