#!/bin/bash
# Issue-side and memory-pipe counters of the sweeps over the LAST `LAST` steps of the 100-step headline run
# (dense floor cells, pressure on), one rocprofv3 --pmc pass per group.  usage: gpu_pmc_late.sh TAG [LAST=20] [bench args]
set -o pipefail
TAG=$1; LAST=${2:-20}; shift; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
cd /tmp
i=0
for ctrs in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VMEM SQ_INSTS_LDS" \
            "SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAVES SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_ANY" \
            "TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum" "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
            "TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum" \
            "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $ctrs --output-format csv -d $OUT/pmc$i -o pmc -- python3 $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 1 --cpu-steps 0 --no-fast-leg --no-linked-leg --no-count-replay --no-extra-legs "$@" > $OUT/pmc$i.json 2> $OUT/pmc$i.err
  echo "pass $i ($ctrs) exit $?"
done
cd $GRAFT_REPO_ROOT
python3 scripts/pmc_summary.py $OUT $LAST > $OUT/pmc_summary.csv; grep -E "density_mask|k_force_" $OUT/pmc_summary.csv
find $OUT -name "*counter_collection.csv" -size +4M -delete
