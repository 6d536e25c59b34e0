#!/bin/bash
# Kernel-trace averages over the last K dispatches of every kernel of one bench command.
# usage: scripts/gpu_ktrace.sh TAG K [bench args]
set -o pipefail
TAG=$1; K=$2; shift 2
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/prof -o bench -- python3 $GRAFT_REPO_ROOT/bench.py --steps $K --cpu-steps 0 --no-linked-leg --no-fast-leg --no-count-replay --no-extra-legs "$@" > $OUT/prof_bench.json 2> $OUT/prof.err; echo "rocprof exit $?"
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
cat $OUT/kernel_stats_timed.csv
