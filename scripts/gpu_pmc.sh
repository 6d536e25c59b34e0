#!/bin/bash
# PMC passes (each in its own rocprofv3 run, counters only) for the bench command.
# usage: scripts/gpu_pmc.sh TAG [bench args]
set -o pipefail
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
cd /tmp
i=0
for ctrs in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAVES SQ_ACTIVE_INST_ANY" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  echo "== pass $i: $ctrs"
  timeout -k 10 300 rocprofv3 --pmc $ctrs --output-format csv -d $OUT/pmc$i -o pmc -- python3 $GRAFT_REPO_ROOT/bench.py --steps ${PMC_STEPS:-10} --warmup 1 --cpu-steps 0 "$@" > $OUT/pmc$i.json 2> $OUT/pmc$i.err
  echo "pass $i exit $?"
done
cd $GRAFT_REPO_ROOT
python3 scripts/pmc_summary.py $OUT > $OUT/pmc_summary.csv; cat $OUT/pmc_summary.csv
# drop the bulky per-dispatch files from what gets merged back
find $OUT -name "*counter_collection.csv" -size +8M -delete
