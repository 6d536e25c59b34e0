#!/bin/bash
set -o pipefail
OUT=gpurun_out/r03_call10; mkdir -p $OUT; export TMPDIR=/tmp
for rep in 1 2; do
BENCH_ARGS="-n 262144 --steps 100 --warmup 5 --no-extra-legs" scripts/gpu_bench_variants.sh r03_call10/n262k main r02 | tee -a $OUT/summary.txt
done
SPH_BENCH_STEP_WALLS=1 python bench.py -n 262144 --steps 40 --warmup 5 --no-extra-legs --cpu-steps 0 --no-linked-leg --no-fast-leg --no-count-replay 2>&1 >/dev/null | grep "step walls" | tee -a $OUT/summary.txt
