#!/bin/bash
set -o pipefail
OUT=gpurun_out/r03_call38; mkdir -p $OUT; export TMPDIR=/tmp
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > $OUT/pytest.txt 2>&1; echo "pytest exit $?" | tee -a $OUT/summary.txt
tail -2 $OUT/pytest.txt | tee -a $OUT/summary.txt
for v in main qhits nosort; do
  if [ "$v" = main ]; then unset SPH_LIB_PATH; else export SPH_LIB_PATH=$PWD/cudafluidsimulator_amd/variants/libsph_hip_$v.so; fi
  timeout -k 10 120 python scripts/studies/kernel_ms.py 16777216 10 2>&1 | tail -1 | tee -a $OUT/summary.txt
done
unset SPH_LIB_PATH
BENCH_ARGS="--steps 100 --warmup 5 --no-extra-legs" scripts/gpu_bench_variants.sh r03_call38/k100 main qhits nosort main qhits | tee -a $OUT/summary.txt
