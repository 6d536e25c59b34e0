#!/bin/bash
# Issue-side and memory-pipe counters of the force sweep for one study command (default: the first 6 steps of
# n = 16,777,216 -i random, single domain), per library variant.  usage: gpu_pmc_study.sh TAG variant [n steps init]
set -o pipefail
TAG=$1; V=$2; N=${3:-16777216}; STEPS=${4:-6}; INIT=${5:-random}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG/$V; mkdir -p $OUT; export TMPDIR=/tmp
if [ "$V" = main ]; then unset SPH_LIB_PATH; else export SPH_LIB_PATH=$GRAFT_REPO_ROOT/cudafluidsimulator_amd/variants/libsph_hip_$V.so; fi
cd /tmp
i=0
for ctrs in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_INSTS_VMEM SQ_INSTS_LDS SQ_INSTS_SALU" \
            "TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum" "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
            "TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum" \
            "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $ctrs --output-format csv -d $OUT/pmc$i -o pmc -- python3 $GRAFT_REPO_ROOT/scripts/studies/kernel_ms.py $N $STEPS $INIT > $OUT/pmc$i.txt 2> $OUT/pmc$i.err
  echo "$V pass $i exit $?"
done
cd $GRAFT_REPO_ROOT
python3 scripts/pmc_summary.py $OUT $STEPS > $OUT/pmc_summary.csv; grep -E "k_force_" $OUT/pmc_summary.csv | cut -d, -f2- 
find $OUT -name "*counter_collection.csv" -size +4M -delete
