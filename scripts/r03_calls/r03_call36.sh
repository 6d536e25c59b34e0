#!/bin/bash
set -o pipefail
OUT=gpurun_out/r03_call36; mkdir -p $OUT; export TMPDIR=/tmp
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_properties.py tests/test_slab.py -m gpu -x -q > $OUT/pytest.txt 2>&1; echo "pytest exit $?" | tee -a $OUT/summary.txt
tail -3 $OUT/pytest.txt | tee -a $OUT/summary.txt
for v in main nosort; do
  if [ "$v" = main ]; then unset SPH_LIB_PATH; else export SPH_LIB_PATH=$PWD/cudafluidsimulator_amd/variants/libsph_hip_$v.so; fi
  timeout -k 10 120 python scripts/studies/kernel_ms.py 16777216 10 2>&1 | tail -1 | tee -a $OUT/summary.txt
  timeout -k 10 120 python scripts/studies/kernel_ms.py 4194304 20 2>&1 | tail -1 | tee -a $OUT/summary.txt
done
unset SPH_LIB_PATH
BENCH_ARGS="--steps 100 --warmup 5 --no-extra-legs" scripts/gpu_bench_variants.sh r03_call36/k100 main nosort main | tee -a $OUT/summary.txt
timeout -k 10 120 python scripts/mgpu_loopback_study.py 4194304 20 random 2,4,8 $OUT/mgpu_4m.json 2>&1 | tee -a $OUT/summary.txt
