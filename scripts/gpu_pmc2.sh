#!/bin/bash
# Extra SQ counter passes (latency / occupancy / FIFO) for the sweeps.
set -o pipefail
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
cd /tmp
i=0
for ctrs in "SQ_LEVEL_WAVES SQ_BUSY_CU_CYCLES SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAIT_INST_LDS SQ_CYCLES" "SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_IFETCH SQ_IFETCH_LEVEL SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_BRANCH SQ_ACTIVE_INST_SCA" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $ctrs --output-format csv -d $OUT/pmc$i -o pmc -- python3 $GRAFT_REPO_ROOT/bench.py --steps ${PMC_STEPS:-10} --warmup 1 --cpu-steps 0 "$@" > $OUT/pmc$i.json 2> $OUT/pmc$i.err
  echo "pass $i exit $?"
done
cd $GRAFT_REPO_ROOT
python3 scripts/pmc_summary.py $OUT > $OUT/pmc_summary.csv; grep -E "^k_density_lds|^k_force_lds" $OUT/pmc_summary.csv
find $OUT -name "*counter_collection.csv" -size +4M -delete
