#!/bin/bash
set -o pipefail
OUT=gpurun_out/r03_call28; mkdir -p $OUT; export TMPDIR=/tmp
BENCH_ARGS="--steps 100 --warmup 5 --no-extra-legs" scripts/gpu_bench_variants.sh r03_call28/k100 main pad20 pad28 pad41 main | tee -a $OUT/summary.txt
