#!/bin/bash
set -o pipefail
OUT=gpurun_out/r03_call34; mkdir -p $OUT; export TMPDIR=/tmp
for v in main nosort s4 s5 s6; do
  if [ "$v" = main ]; then unset SPH_LIB_PATH; else export SPH_LIB_PATH=$PWD/cudafluidsimulator_amd/variants/libsph_hip_$v.so; fi
  timeout -k 10 120 python scripts/studies/kernel_ms.py 16777216 10 2>&1 | tail -1 | tee -a $OUT/summary.txt
  timeout -k 10 120 python scripts/studies/kernel_ms.py 67108864 3 grid 2>&1 | tail -1 | tee -a $OUT/summary.txt
done
unset SPH_LIB_PATH
BENCH_ARGS="--steps 100 --warmup 5 --no-extra-legs" scripts/gpu_bench_variants.sh r03_call34/k100 main nosort s4 s5 s6 | tee -a $OUT/summary.txt
