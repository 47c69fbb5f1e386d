#!/bin/bash
# Real timeline of the replayed iteration: rocprofv3 --kernel-trace timestamps of a short bench run; per kernel name the mean
# duration and the mean start-to-next-start interval (duration + the gap behind it) over the steady-state iterations.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
TAG=${1:-r02}
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/prof_t
rocprofv3 --kernel-trace --output-format csv -d $OUT/prof_t -o t -- python3 $ROOT/bench.py --algo ${ALGO:-sac} --steps 300 --warmup 50 --no-variant --no-cpu-baseline --no-roofline > $OUT/prof_t.json 2> $OUT/prof_t.err
f=$(find $OUT/prof_t -name "*kernel_trace.csv" | head -1)
python3 - <<PY
import csv, collections, json, statistics
rows = list(csv.DictReader(open("$f")))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
name = lambda r: r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:70]
# steady state: the last 60 % of the dispatches
rows = rows[int(0.4 * len(rows)):]
dur, itv = collections.defaultdict(list), collections.defaultdict(list)
for a, b in zip(rows, rows[1:]):
    s, e, ns = int(a["Start_Timestamp"]), int(a["End_Timestamp"]), int(b["Start_Timestamp"])
    if ns - s < 200000:  # not across a host gap
        dur[name(a)].append(e - s)
        itv[name(a)].append(ns - s)
iters = max(len(v) for k, v in dur.items() if "collect_step" in k or "rollout_step" in k)
out = []
for k in dur:
    out.append((sum(itv[k]) / iters / 1e3, k, len(dur[k]) / iters, statistics.mean(dur[k]) / 1e3, statistics.mean(itv[k]) / 1e3))
out.sort(reverse=True)
tot = sum(o[0] for o in out)
print(f"iteration (sum of intervals) {tot:.1f} us under the profiler; bench says", json.load(open("$OUT/prof_t.json"))["ms_per_step"], "ms")
print(f"{'kernel':70s} {'per it':>6s} {'dur us':>7s} {'intv us':>7s} {'us/it':>7s}")
for us_it, k, n, d, i in out:
    if n >= 0.3:
        print(f"{k:70s} {n:6.2f} {d:7.2f} {i:7.2f} {us_it:7.2f}")
json.dump([dict(kernel=k, per_iteration=n, duration_us=d, interval_us=i, us_per_iteration=u) for u, k, n, d, i in out if n >= 0.3],
          open("$OUT/${TAG}_graph_timeline.json", "w"), indent=1)
PY
rm -rf $OUT/prof_t
