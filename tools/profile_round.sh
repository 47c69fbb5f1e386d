#!/bin/bash
# Round measurements on the GPU box: the default bench line and rocprofv3 kernel-trace summaries of short SAC / TD3 / MADDPG
# runs (same command, fewer steps). Outputs under gpurun_out/; copy what is to be judged into profiles/.
set -e
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
TAG=${1:-final}
[ -n "$SKIP_BENCH" ] || python $ROOT/bench.py > $OUT/bench_$TAG.json 2> $OUT/bench_$TAG.err
cd /tmp && export TMPDIR=/tmp
for algo in sac td3 maddpg; do
  extra=""; [ $algo = maddpg ] && extra="--n-envs 1024"
  rm -rf $OUT/prof_$algo
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$algo -o $algo -- python3 $ROOT/bench.py --algo $algo $extra --steps 150 --warmup 50 --no-variant --no-cpu-baseline > $OUT/prof_$algo.json 2> $OUT/prof_$algo.err
  find $OUT/prof_$algo -name "*kernel_stats.csv" -exec cp {} $OUT/${TAG}_${algo}_kernel_stats.csv \;
  rm -rf $OUT/prof_$algo
done
python3 - <<PY
import json
d = json.load(open("$OUT/bench_$TAG.json"))
print(d["value"], d["ms_per_step"], d["roofline"], d.get("speedup_vs_cpu_port"))
PY
