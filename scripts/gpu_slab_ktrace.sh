#!/bin/bash
# Timeline of one slab step (kernel-trace): usage: scripts/gpu_slab_ktrace.sh TAG [N] [steps]
set -o pipefail
TAG=$1; N=${2:-8}; STEPS=${3:-12}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/prof -o slab -- python3 $GRAFT_REPO_ROOT/scripts/studies/slab_step_trace.py $N $STEPS > $OUT/run.txt 2> $OUT/prof.err; echo "rocprof exit $?"
python3 - <<PY
import csv, glob
rows = []
for f in glob.glob("$OUT/prof/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f, newline="")):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:60]))
rows.sort()
# the last slab step = from the last k_partition_count (or the last first-pass histogram) to the end
starts = [i for i, r in enumerate(rows) if "partition_count" in r[2] or "k_classify" in r[2]]
i0 = starts[-1] if starts else max(0, len(rows) - 16)
t0 = rows[i0][0]; prev = None
with open("$OUT/slab_step_timeline.txt", "w") as out:
    for s, e, k in rows[i0:]:
        line = "%8.1f us  +%6.1f us  gap %6.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev) / 1e3 if prev else 0.0, k)
        print(line); out.write(line + "\n")
        prev = e
    print("step span %.1f us, kernels %.1f us" % ((rows[-1][1] - t0) / 1e3, sum(e - s for s, e, _ in rows[i0:]) / 1e3))
PY
find $OUT/prof -name "*kernel_trace.csv" -delete
