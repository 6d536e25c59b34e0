#!/bin/bash
# soak to step 400 (list vs lds families, sha256 at checkpoints), then the round's profile set on the final build
set -o pipefail
OUT=gpurun_out/r03_call40; mkdir -p $OUT; export TMPDIR=/tmp
bash scripts/gpu_profile_round.sh r03_prof_k20 20 5 2>&1 | tee -a $OUT/summary.txt
bash scripts/gpu_profile_round.sh r03_prof_k100 100 5 2>&1 | tee -a $OUT/summary.txt
bash scripts/gpu_pmc_late.sh r03_pmc_late 20 2>&1 | tail -60 > $OUT/pmc_late.txt
