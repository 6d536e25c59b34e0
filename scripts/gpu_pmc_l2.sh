#!/bin/bash
# L2 behaviour of the sweeps: FETCH_SIZE and TCC hit/miss, one rocprofv3 --pmc run each.
# usage: scripts/gpu_pmc_l2.sh TAG [bench args]
set -o pipefail
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
cd /tmp
i=0
for ctrs in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $ctrs --output-format csv -d $OUT/pmc$i -o pmc -- python3 $GRAFT_REPO_ROOT/bench.py --steps ${PMC_STEPS:-100} --warmup 1 --cpu-steps 0 --no-fast-leg --no-linked-leg --no-count-replay "$@" > $OUT/pmc$i.json 2> $OUT/pmc$i.err
  echo "pass $i exit $?"
done
cd $GRAFT_REPO_ROOT
python3 scripts/pmc_summary.py $OUT > $OUT/pmc_summary.csv; grep -i "force_list\|density_mask" $OUT/pmc_summary.csv
find $OUT -name "*counter_collection.csv" -size +4M -delete
