#!/bin/bash
# per-iteration launch census of a short SAC bench run under rocprofv3 (kernel name, calls, average us), with optional env knobs
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/prof_x
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_x -o sac -- python3 $ROOT/bench.py --algo ${ALGO:-sac} ${EXTRA:-} --graph-unroll 1 --steps 300 --warmup 50 --no-variant --no-cpu-baseline --no-roofline > $OUT/prof_x.json 2> $OUT/prof_x.err
f=$(find $OUT/prof_x -name "*kernel_stats.csv" | head -1)
python3 - <<PY
import csv, json
rows = list(csv.DictReader(open("$f")))
iters = max(int(r["Calls"]) for r in rows if "collect_step_kernel" in r["Name"] or "rollout_step_kernel" in r["Name"])
keep = [r for r in rows if int(r["Calls"]) >= 0.4 * iters]
print("iterations", iters, "launches per iteration", round(sum(int(r["Calls"]) for r in keep) / iters, 2), "ms_per_step (under rocprof)",
      json.load(open("$OUT/prof_x.json"))["ms_per_step"])
for r in keep:
    print(r["Name"][:90].ljust(90), round(int(r["Calls"]) / iters, 2), round(float(r["AverageNs"]) / 1e3, 2))
PY
rm -rf $OUT/prof_x
