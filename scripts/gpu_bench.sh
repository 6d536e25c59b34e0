#!/bin/bash
# One gpurun call: bench.py (N=1) and rocprofv3 kernel-trace of the same command.
# usage: scripts/gpu_bench.sh [tag] [extra bench args...]
set -o pipefail
TAG=${1:-r1}; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo "== bench lds" | tee $OUT/status.txt
timeout -k 10 400 python bench.py --steps 100 --warmup 5 "$@" > $OUT/bench_lds.json 2> $OUT/bench_lds.err
echo "bench exit $?" | tee -a $OUT/status.txt; cat $OUT/bench_lds.json
echo "== bench direct" | tee -a $OUT/status.txt
timeout -k 10 300 python bench.py --steps 100 --warmup 5 --sweep direct --cpu-steps 0 "$@" > $OUT/bench_direct.json 2> $OUT/bench_direct.err
echo "bench exit $?" | tee -a $OUT/status.txt; cat $OUT/bench_direct.json
echo "== bench 262144" | tee -a $OUT/status.txt
timeout -k 10 200 python bench.py --steps 100 --warmup 5 -n 262144 --cpu-steps 0 > $OUT/bench_262k.json 2> $OUT/bench_262k.err
cat $OUT/bench_262k.json
echo "== rocprofv3 kernel trace" | tee -a $OUT/status.txt
(cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof -o bench -- python3 $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 5 --cpu-steps 0 "$@" > $GRAFT_REPO_ROOT/$OUT/prof_bench.json 2> $GRAFT_REPO_ROOT/$OUT/prof.err)
echo "rocprof exit $?" | tee -a $OUT/status.txt
find $OUT/prof -name "*stats*" | head; for f in $(find $OUT/prof -name "*kernel_stats.csv"); do head -20 $f; done
# keep the merged output small: drop the per-dispatch trace, keep the stats
find $OUT/prof -name "*kernel_trace.csv" -size +20M -delete
