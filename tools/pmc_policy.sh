#!/bin/bash
# HBM traffic of the policy kernel from the PMC counters (separate passes, as the MI355X guide prescribes): FETCH_SIZE (KiB,
# doubled on gfx950) and WRITE_SIZE (KiB) per dispatch of policy_rows_fwd_kernel in tools/microbench_policy.py (median over dispatches).
# Other kernels: BENCH=tools/microbench_rollout.py KERNEL=rollout_step NAME=rollout tools/pmc_policy.sh -> gpurun_out/rollout_pmc.json
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
BENCH=${BENCH:-tools/microbench_policy.py}
KERNEL=${KERNEL:-policy_rows}
NAME=${NAME:-policy}
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $OUT/pmc_$c
  rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_$c -o p -- python3 $ROOT/$BENCH > /dev/null 2> $OUT/pmc_$c.err
done
python3 - <<PY
import csv, glob, json, statistics
res = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob("$OUT/pmc_%s/**/*counter_collection.csv" % c, recursive=True)[0]
    v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "$KERNEL" in r["Kernel_Name"] and r["Counter_Name"] == c]
    res[c + "_KiB"] = round(statistics.median(v), 2)
    res["dispatches"] = len(v)
print(json.dumps(res))
res["traffic_bytes"] = int((2 * res["FETCH_SIZE_KiB"] + res["WRITE_SIZE_KiB"]) * 1024)  # gfx950: FETCH_SIZE counts half the bytes
print(json.dumps(res))
json.dump(res, open("$OUT/${NAME}_pmc.json", "w"))
PY
rm -rf $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE
