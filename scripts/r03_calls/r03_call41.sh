#!/bin/bash
# density sweep: select-free trips when no lane is in its partial trip
set -o pipefail
OUT=gpurun_out/r03_call41; mkdir -p $OUT; export TMPDIR=/tmp
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_properties.py tests/test_slab.py -m gpu -x -q > $OUT/pytest.txt 2>&1; echo "pytest exit $?" | tee -a $OUT/summary.txt
tail -2 $OUT/pytest.txt | tee -a $OUT/summary.txt
BENCH_ARGS="--steps 100 --warmup 5 --no-extra-legs" scripts/gpu_bench_variants.sh r03_call41/k100 main slowtrip main slowtrip | tee -a $OUT/summary.txt
for v in main slowtrip; do
  if [ "$v" = main ]; then unset SPH_LIB_PATH; else export SPH_LIB_PATH=$PWD/cudafluidsimulator_amd/variants/libsph_hip_$v.so; fi
  timeout -k 10 120 python scripts/studies/kernel_ms.py 16777216 10 2>&1 | tail -1 | tee -a $OUT/summary.txt
  timeout -k 10 120 python scripts/studies/kernel_ms.py 262144 100 2>&1 | tail -1 | tee -a $OUT/summary.txt
done
