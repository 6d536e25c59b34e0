#!/bin/bash
set -o pipefail
OUT=gpurun_out/r03_call42; mkdir -p $OUT; export TMPDIR=/tmp
for v in main nodeal main nodeal; do
  if [ "$v" = main ]; then unset SPH_LIB_PATH; else export SPH_LIB_PATH=$PWD/cudafluidsimulator_amd/variants/libsph_hip_$v.so; fi
  timeout -k 10 120 python scripts/studies/kernel_ms.py 262144 100 2>&1 | tail -1 | tee -a $OUT/summary.txt
  timeout -k 10 120 python scripts/studies/kernel_ms.py 262144 300 2>&1 | tail -1 | tee -a $OUT/summary.txt
done
