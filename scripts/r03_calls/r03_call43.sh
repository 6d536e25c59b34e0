#!/bin/bash
set -o pipefail
OUT=gpurun_out/r03_call43; mkdir -p $OUT; export TMPDIR=/tmp
BENCH_ARGS="--steps 100 --warmup 5 --no-extra-legs" scripts/gpu_bench_variants.sh r03_call43/k100 main ldsonly | tee -a $OUT/summary.txt
