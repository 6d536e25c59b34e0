#!/bin/bash
set -o pipefail
OUT=gpurun_out/r03_call4; mkdir -p $OUT; export TMPDIR=/tmp
timeout -k 10 700 python -m pytest tests -m gpu -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/summary.txt
tail -15 $OUT/pytest.log | tee -a $OUT/summary.txt
timeout -k 10 200 python scripts/studies/per_step_profile.py --count --every 5 > $OUT/profile.txt 2>&1 && grep "^step" $OUT/profile.txt | tee -a $OUT/summary.txt
BENCH_ARGS="--steps 100 --warmup 5 --no-extra-legs" scripts/gpu_bench_variants.sh r03_call4/k100 main | tee -a $OUT/summary.txt
BENCH_ARGS="--steps 20 --warmup 5 --no-extra-legs" scripts/gpu_bench_variants.sh r03_call4/k20 main | tee -a $OUT/summary.txt
