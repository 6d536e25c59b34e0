#!/bin/bash
# One round's evidence for the bench line, all from the SAME bench command line
# (bench.py --steps K --warmup W): the plain bench line, the rocprofv3 kernel-trace
# stats, and the HBM traffic counters (FETCH_SIZE / WRITE_SIZE in separate --pmc
# passes, as MI355X_MICROARCH.md prescribes).  Summaries land in gpurun_out/TAG/;
# copy what should be judged into profiles/.
# usage: scripts/gpu_profile_round.sh TAG K W [extra bench args]
set -o pipefail
TAG=$1; K=$2; W=$3; shift 3
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout -k 10 400 python bench.py --steps $K --warmup $W "$@" > $OUT/bench.json 2> $OUT/bench.err; echo "bench exit $?"
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o bench -- python3 $GRAFT_REPO_ROOT/bench.py --steps $K --warmup $W --cpu-steps 0 --no-linked-leg --no-fast-leg --no-count-replay --no-extra-legs "$@" > $OUT/prof_bench.json 2> $OUT/prof.err; echo "rocprof exit $?"
# averages over the TIMED steps only (the last K dispatches of each kernel: settle and warm-up
# steps come first) -- the figure bench.py's HIP-event average must agree with
python3 - <<PY
import csv, glob
from collections import defaultdict
K = $K
v = defaultdict(list)
for f in glob.glob("$OUT/prof/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f, newline="")):
        v[r["Kernel_Name"].split("(")[0]].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
w = csv.writer(open("$OUT/kernel_stats_timed.csv", "w", newline=""))
w.writerow(["kernel", "dispatches", "avg_ns_all", "avg_ns_last_%d" % K])
for k, d in sorted(v.items(), key=lambda kv: -sum(x for _, x in kv[1])):
    d.sort()
    t = d[-K:] if len(d) >= K else d
    w.writerow([k, len(d), "%.1f" % (sum(x for _, x in d) / len(d)), "%.1f" % (sum(x for _, x in t) / len(t))])
PY
find $OUT/prof -name "*kernel_trace.csv" -delete
i=0
for ctrs in "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $ctrs --output-format csv -d $OUT/pmc$i -o pmc -- python3 $GRAFT_REPO_ROOT/bench.py --steps $K --warmup $W --cpu-steps 0 --no-linked-leg --no-fast-leg --no-count-replay --no-extra-legs "$@" > $OUT/pmc$i.json 2> $OUT/pmc$i.err; echo "pmc pass $i exit $?"
done
cd $GRAFT_REPO_ROOT
python3 scripts/pmc_summary.py $OUT $K > $OUT/pmc_summary.csv
find $OUT -name "*counter_collection.csv" -size +2M -delete
find $OUT -name "*_agent_info.csv" -delete
python3 - <<PY
import json, csv
d = json.load(open("$OUT/bench.json")); k = d["kernel_ms_per_step"]; r = d["roofline"]
print("value %.3e ms/step %.3f | density %.3f force %.3f sort %.3f gather %.3f hash %.3f" % (d["value"], d["ms_per_step"], k["density"], k["force_integrate"], k["sort"], k["gather_cells"], k["hash"]))
print({x: r.get(x) for x in ("frac", "valu_frac", "force_valu_frac", "force_hit_fraction", "traffic")})
print(d.get("cpu_baseline", {}).get("sample"))
for row in csv.DictReader(open("$OUT/pmc_summary.csv")):
    if "density_mask" in row["kernel"] or "k_force_" in row["kernel"]:
        print(row["kernel"][:40], row["counter"], row["avg_per_dispatch"], row["dispatches"], "timed steps only:", row.get("avg_last_%d_dispatches" % $K))
PY
head -8 $OUT/prof/bench_kernel_stats.csv | cut -c1-160
