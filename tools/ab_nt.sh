#!/bin/bash
# A/B: non-temporal ring stores (tools/ab/lib_nt1.so, built with `make NT=1`) vs plain stores, interleaved rounds.
for round in 1 2 3 4; do
  for lib in "" "tools/ab/lib_nt1.so"; do
    for n in 4194304 4096; do
      if [ -n "$lib" ]; then export CSTR_LIB_PATH=$PWD/$lib; else unset CSTR_LIB_PATH; fi
      python tools/microbench_collect.py $n 40 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('nt=${lib:+1}', d['n_envs'], d['launch_us'], d['achieved'])"
    done
  done
done
unset CSTR_LIB_PATH
for lib in "" "tools/ab/lib_nt1.so"; do
  if [ -n "$lib" ]; then export CSTR_LIB_PATH=$PWD/$lib; else unset CSTR_LIB_PATH; fi
  python bench.py --no-cpu-baseline --no-roofline --steps 300 2>/dev/null | cut -c1-200
done
