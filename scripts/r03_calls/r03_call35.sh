#!/bin/bash
set -o pipefail
OUT=gpurun_out/r03_call35; mkdir -p $OUT; export TMPDIR=/tmp
for v in main nosort ident k128 k64s; do
  if [ "$v" = main ]; then unset SPH_LIB_PATH; else export SPH_LIB_PATH=$PWD/cudafluidsimulator_amd/variants/libsph_hip_$v.so; fi
  timeout -k 10 120 python scripts/studies/kernel_ms.py 16777216 10 2>&1 | tail -1 | tee -a $OUT/summary.txt
done
