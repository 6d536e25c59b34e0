#!/bin/bash
set -o pipefail
OUT=gpurun_out/r03_call8; mkdir -p $OUT; export TMPDIR=/tmp
BENCH_ARGS="--steps 100 --warmup 5 --no-extra-legs" scripts/gpu_bench_variants.sh r03_call8/k100 main win192 win224 win256 win320 win384 | tee -a $OUT/summary.txt
