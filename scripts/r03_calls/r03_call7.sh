#!/bin/bash
set -o pipefail
OUT=gpurun_out/r03_call7; mkdir -p $OUT; export TMPDIR=/tmp
BENCH_ARGS="--steps 100 --warmup 5 --no-extra-legs" scripts/gpu_bench_variants.sh r03_call7/k100 main win96 win128 win192 k2t128 k1w5 k1w7 cap320 cap448 main | tee -a $OUT/summary.txt
