#!/bin/bash
set -o pipefail
OUT=gpurun_out/r03_call25; mkdir -p $OUT; export TMPDIR=/tmp
BENCH_ARGS="--steps 100 --warmup 5 --no-extra-legs" scripts/gpu_bench_variants.sh r03_call25/k100 main sortl sortl512 sortls1 sortls2 sortl128 sortl | tee -a $OUT/summary.txt
