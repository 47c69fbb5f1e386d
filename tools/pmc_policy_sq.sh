#!/bin/bash
# SQ counters of the policy kernel: matrix-core busy cycles, LDS bank conflicts, wait buckets -- separate
# rocprofv3 --pmc passes over tools/microbench_policy.py (program directly after `--`), median per dispatch of the policy kernel.
# Usage: tools/pmc_policy_sq.sh TAG  -> gpurun_out/TAG_policy_sq_pmc.json (+ the available-counter list once)
# Other kernels: BENCH=tools/microbench_rollout.py KERNEL=rollout_step NAME=rollout tools/pmc_policy_sq.sh TAG -> TAG_rollout_sq_pmc.json
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
TAG=${1:-r02}
BENCH=${BENCH:-tools/microbench_policy.py}
KERNEL=${KERNEL:-policy_rows}
NAME=${NAME:-policy}
cd /tmp && export TMPDIR=/tmp
[ -f $OUT/rocprof_counters_gfx950.txt ] || rocprofv3 -L > $OUT/rocprof_counters_gfx950.txt 2>&1
PASSES=(
 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES"
 "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_SALU"
 "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS"
 "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU"
 "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum"
 "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"
)
i=0
for p in "${PASSES[@]}"; do
  rm -rf $OUT/sq_$i
  rocprofv3 --pmc $p --output-format csv -d $OUT/sq_$i -o p -- python3 $ROOT/$BENCH > /dev/null 2> $OUT/sq_$i.err || echo "pass $i failed: $p"
  i=$((i+1))
done
python3 - <<PY
import csv, glob, json, statistics
res = {}
for d in sorted(glob.glob("$OUT/sq_*/")):
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        rows = [r for r in csv.DictReader(open(f)) if "$KERNEL" in r["Kernel_Name"]]
        for name in sorted({r["Counter_Name"] for r in rows}):
            v = [float(r["Counter_Value"]) for r in rows if r["Counter_Name"] == name]
            res[name] = dict(median=statistics.median(v), dispatches=len(v))
print(json.dumps(res, indent=1))
json.dump(res, open("$OUT/${TAG}_${NAME}_sq_pmc.json", "w"), indent=1)
PY
rm -rf $OUT/sq_[0-9]*
