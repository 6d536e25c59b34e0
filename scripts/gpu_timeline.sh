#!/bin/bash
# Where a step's wall time goes: kernel-trace timestamps of the last K steps (compute kernels vs the read-back's
# blit kernel), gaps between consecutive kernels and between steps.  usage: scripts/gpu_timeline.sh TAG K [bench args]
set -o pipefail
TAG=$1; K=$2; shift 2
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/prof -o bench -- python3 $GRAFT_REPO_ROOT/bench.py --steps $K --cpu-steps 0 --no-linked-leg --no-fast-leg --no-count-replay --no-extra-legs "$@" > $OUT/prof_bench.json 2> $OUT/prof.err; echo "rocprof exit $?"
python3 - <<PY
import csv, glob
K = $K
rows = []
for f in glob.glob("$OUT/prof/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f, newline="")):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0]))
rows.sort()
comp = [r for r in rows if "rocclr" not in r[2]]
copy = [r for r in rows if "copyBuffer" in r[2]]
# steps: a step starts with k_radix_hist<..., true, ...>
starts = [i for i, r in enumerate(comp) if "k_radix_hist" in r[2] and "true" in r[2]]
starts = starts[-K:]
out = []
for a, b in zip(starts, starts[1:] + [len(comp)]):
    ks = comp[a:b]
    busy = sum(e - s for s, e, _ in ks)
    span = ks[-1][1] - ks[0][0]
    gaps = [ks[i + 1][0] - ks[i][1] for i in range(len(ks) - 1)]
    out.append((ks[0][0], ks[-1][1], busy, span, sum(g for g in gaps if g > 0), len(ks)))
print("step: kernels_busy_us span_us gaps_inside_us n_kernels | gap_to_next_step_us | period_us")
for i, o in enumerate(out):
    nxt = out[i + 1][0] - o[1] if i + 1 < len(out) else 0
    per = out[i + 1][0] - o[0] if i + 1 < len(out) else 0
    print("%2d: %7.1f %7.1f %6.1f %d | %6.1f | %7.1f" % (i + 1, o[2] / 1e3, o[3] / 1e3, o[4] / 1e3, o[5], nxt / 1e3, per / 1e3))
cp = copy[-K:]
print("copies: dur_us", " ".join("%.0f" % ((e - s) / 1e3) for s, e, _ in cp))
print("copy start - previous copy end (us):", " ".join("%.0f" % ((cp[i + 1][0] - cp[i][1]) / 1e3) for i in range(len(cp) - 1)))
PY
find $OUT/prof -name "*kernel_trace.csv" -delete
