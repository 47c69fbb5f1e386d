#!/bin/bash
# Kernel launches per learn() iteration = (total launches of a 350-step run - total of a 150-step run) / 200, from rocprofv3
# --kernel-trace --stats summaries of the same bench command.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
for algo in sac td3 maddpg; do
  extra=""; [ $algo = maddpg ] && extra="--n-envs 1024"
  for steps in 150 350; do
    rm -rf $OUT/cnt
    rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/cnt -o c -- python3 $ROOT/bench.py --algo $algo $extra --steps $steps --warmup 50 --no-variant --no-cpu-baseline > /dev/null 2> $OUT/cnt.err
    f=$(find $OUT/cnt -name "*kernel_stats.csv" | head -1)
    python3 -c "import csv,sys; print('$algo', $steps, sum(int(r['Calls']) for r in csv.DictReader(open('$f'))))"
  done
  rm -rf $OUT/cnt
done
