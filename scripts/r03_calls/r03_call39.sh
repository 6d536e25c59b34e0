#!/bin/bash
# refactored dealt force kernel (main = 256-row groups) and its two-pass "snake" form (512-row groups)
set -o pipefail
OUT=gpurun_out/r03_call39; mkdir -p $OUT; export TMPDIR=/tmp
for v in main snake; do
  if [ "$v" = main ]; then unset SPH_LIB_PATH; else export SPH_LIB_PATH=$PWD/cudafluidsimulator_amd/variants/libsph_hip_$v.so; fi
  timeout -k 10 500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_properties.py tests/test_slab.py -m gpu -x -q > $OUT/pytest_$v.txt 2>&1; echo "pytest $v exit $?" | tee -a $OUT/summary.txt
  tail -2 $OUT/pytest_$v.txt | tee -a $OUT/summary.txt
done
for v in main snake nodeal; do
  if [ "$v" = main ]; then unset SPH_LIB_PATH; else export SPH_LIB_PATH=$PWD/cudafluidsimulator_amd/variants/libsph_hip_$v.so; fi
  timeout -k 10 120 python scripts/studies/kernel_ms.py 16777216 10 2>&1 | tail -1 | tee -a $OUT/summary.txt
done
unset SPH_LIB_PATH
BENCH_ARGS="--steps 100 --warmup 5 --no-extra-legs" scripts/gpu_bench_variants.sh r03_call39/k100 main snake nodeal main snake | tee -a $OUT/summary.txt
