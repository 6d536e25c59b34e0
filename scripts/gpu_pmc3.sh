#!/bin/bash
# Memory-pipeline counters (TA / TCP / TCC) for a bench command.  usage: gpu_pmc3.sh TAG [bench args]
set -o pipefail
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
cd /tmp
i=0
for ctrs in "TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum" "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" "TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum" "TCP_TAGRAM0_REQ_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum" "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_BUSY_sum" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $ctrs --output-format csv -d $OUT/pmc$i -o pmc -- python3 $GRAFT_REPO_ROOT/bench.py --steps ${PMC_STEPS:-10} --warmup 1 --cpu-steps 0 --no-fast-leg --no-linked-leg --no-count-replay "$@" > $OUT/pmc$i.json 2> $OUT/pmc$i.err
  echo "pass $i ($ctrs) exit $?"
done
cd $GRAFT_REPO_ROOT
python3 scripts/pmc_summary.py $OUT ${PMC_STEPS:-10} > $OUT/pmc_summary.csv; grep -E "density|force" $OUT/pmc_summary.csv
find $OUT -name "*counter_collection.csv" -size +4M -delete
