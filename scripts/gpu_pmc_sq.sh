#!/bin/bash
# Issue-side counters of the sweeps (one rocprofv3 --pmc run per group, counters only).
# usage: scripts/gpu_pmc_sq.sh TAG [bench args]
set -o pipefail
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
cd /tmp
i=0
for ctrs in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAVES SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_THREAD_CYCLES_VALU" "GRBM_GUI_ACTIVE SQ_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VMEM SQ_INSTS_FLAT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $ctrs --output-format csv -d $OUT/pmc$i -o pmc -- python3 $GRAFT_REPO_ROOT/bench.py --steps ${PMC_STEPS:-20} --warmup 1 --cpu-steps 0 --no-fast-leg --no-linked-leg --no-count-replay "$@" > $OUT/pmc$i.json 2> $OUT/pmc$i.err
  echo "pass $i exit $?"
done
cd $GRAFT_REPO_ROOT
python3 scripts/pmc_summary.py $OUT ${PMC_STEPS:-20} > $OUT/pmc_summary.csv; grep -i "force_list\|density_mask" $OUT/pmc_summary.csv
find $OUT -name "*counter_collection.csv" -size +4M -delete
