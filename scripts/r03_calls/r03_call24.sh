#!/bin/bash
# sorted-lanes force variant: parity subset under the variant library, then K=100 and K=20 A/B
set -o pipefail
OUT=gpurun_out/r03_call24; mkdir -p $OUT; export TMPDIR=/tmp
SPH_LIB_PATH=$PWD/cudafluidsimulator_amd/variants/libsph_hip_sortl.so timeout -k 10 500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_properties.py tests/test_slab.py -m gpu -x -q > $OUT/pytest_sortl.txt 2>&1; echo "pytest sortl exit $?" | tee -a $OUT/summary.txt
tail -3 $OUT/pytest_sortl.txt | tee -a $OUT/summary.txt
BENCH_ARGS="--steps 100 --warmup 5 --no-extra-legs" scripts/gpu_bench_variants.sh r03_call24/k100 main sortl main sortl | tee -a $OUT/summary.txt
BENCH_ARGS="--steps 20 --warmup 5 --no-extra-legs" scripts/gpu_bench_variants.sh r03_call24/k20 main sortl | tee -a $OUT/summary.txt
