#!/bin/bash
# Sweep of HIP runtime environment knobs over the default bench (results: profiles/r01_notes.md). Run on the GPU box.
run() { # name, env...
  name=$1; shift
  env "$@" timeout -k 10 200 python bench.py --no-variant --no-cpu-baseline --steps 3000 > gpurun_out/es_$name.json 2>gpurun_out/es_$name.err
  python -c "import json; d=json.load(open('gpurun_out/es_$name.json')); print('$name', d['value'], d['ms_per_step'])" 2>&1 | tail -1
}
run base A=1
run devkernarg1 HIP_FORCE_DEV_KERNARG=1
run devkernarg0 HIP_FORCE_DEV_KERNARG=0
run pktcap0 DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
run pktcap1 DEBUG_CLR_GRAPH_PACKET_CAPTURE=1
run optflush0 AMD_OPT_FLUSH=0
run fgs0 ROC_USE_FGS_KERNARG=0
run batch1 DEBUG_HIP_GRAPH_BATCH_SIZE=1
run batch64 DEBUG_HIP_GRAPH_BATCH_SIZE=64
run base2 A=1
