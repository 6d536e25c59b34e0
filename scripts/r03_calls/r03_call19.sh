#!/bin/bash
set -o pipefail
OUT=gpurun_out/r03_call19; mkdir -p $OUT; export TMPDIR=/tmp
for P in 1 0 1 0; do
  export SPH_PIPELINE=$P; echo "== SPH_PIPELINE=$P (SDMA read-back on)" | tee -a $OUT/summary.txt
  BENCH_ARGS="--steps 100 --warmup 5 --no-extra-legs" scripts/gpu_bench_variants.sh r03_call19/k100_$P main | tee -a $OUT/summary.txt
  BENCH_ARGS="--steps 20 --warmup 5 --no-extra-legs" scripts/gpu_bench_variants.sh r03_call19/k20_$P main | tee -a $OUT/summary.txt
done
