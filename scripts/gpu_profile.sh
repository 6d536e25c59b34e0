#!/bin/bash
# Round profile set: bench, kernel-trace stats, HBM PMC passes -> gpurun_out/TAG
set -o pipefail
TAG=$1; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench exit $?"; cat $OUT/bench.json | cut -c1-600
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o bench -- python3 $GRAFT_REPO_ROOT/bench.py --cpu-steps 0 --no-fast-leg > $OUT/prof_bench.json 2> $OUT/prof.err; echo "rocprof exit $?"
rm -f $OUT/prof/bench_kernel_trace.csv
i=0
for ctrs in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $ctrs --output-format csv -d $OUT/pmc$i -o pmc -- python3 $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 1 --cpu-steps 0 --no-fast-leg > $OUT/pmc$i.json 2> $OUT/pmc$i.err; echo "pmc pass $i exit $?"
done
cd $GRAFT_REPO_ROOT
python3 scripts/pmc_summary.py $OUT > $OUT/pmc_summary.csv; cat $OUT/pmc_summary.csv
find $OUT -name "*counter_collection.csv" -size +4M -delete
