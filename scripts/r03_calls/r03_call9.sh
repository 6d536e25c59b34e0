#!/bin/bash
set -o pipefail
OUT=gpurun_out/r03_call9; mkdir -p $OUT; export TMPDIR=/tmp
timeout -k 10 700 python -m pytest tests -m gpu -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/summary.txt
tail -6 $OUT/pytest.log | tee -a $OUT/summary.txt
bash scripts/gpu_profile_round.sh r03_prof_k20 20 5 2>&1 | tee -a $OUT/summary.txt
bash scripts/gpu_profile_round.sh r03_prof_k100 100 5 2>&1 | tee -a $OUT/summary.txt
