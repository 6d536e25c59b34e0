#!/bin/bash
set -o pipefail
OUT=gpurun_out/r03_call13; mkdir -p $OUT; export TMPDIR=/tmp
BENCH_ARGS="--steps 100 --warmup 5 --no-extra-legs" scripts/gpu_bench_variants.sh r03_call13/k100 main rt rt8 main rt rt8 | tee -a $OUT/summary.txt
