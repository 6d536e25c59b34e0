#!/bin/bash
# round 3, GPU call 1: parity suite with the zero-pair filter, per-step profile on/off, K=100/K=20 bench, WG-size variants
set -o pipefail
OUT=gpurun_out/r03_call1; mkdir -p $OUT; export TMPDIR=/tmp
timeout -k 10 420 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/summary.txt
tail -5 $OUT/pytest.log | tee -a $OUT/summary.txt
timeout -k 10 200 python scripts/studies/per_step_profile.py --count --every 5 > $OUT/profile_filter_on.txt 2>&1 && tail -22 $OUT/profile_filter_on.txt | head -21 | tee -a $OUT/summary.txt
SPH_ZERO_PAIR_FILTER=0 timeout -k 10 200 python scripts/studies/per_step_profile.py --every 5 > $OUT/profile_filter_off.txt 2>&1 && tail -22 $OUT/profile_filter_off.txt | head -21 | tee -a $OUT/summary.txt
BENCH_ARGS="--steps 100 --warmup 5" scripts/gpu_bench_variants.sh r03_call1/k100 main k2t256 k2t512 | tee -a $OUT/summary.txt
BENCH_ARGS="--steps 20 --warmup 5" scripts/gpu_bench_variants.sh r03_call1/k20 main | tee -a $OUT/summary.txt
