#!/bin/bash
# A/B of launch shapes for the streaming regime of the collect kernel (N = 2^22); each variant in its own process,
# interleaved rounds. Usage on the GPU box: bash tools/ab_collect.sh
for round in 1 2 3; do
  for v in "256 2048" "256 4096" "256 8192" "256 16384" "512 2048" "512 4096" "1024 1024" "1024 2048" "128 4096" "64 16384"; do
    set -- $v
    CSTR_ENV_BLOCK=$1 CSTR_ENV_GRID_CAP=$2 python tools/microbench_collect.py 4194304 30 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('block $1 cap $2', d['launch_us'], d['achieved'])"
  done
done
