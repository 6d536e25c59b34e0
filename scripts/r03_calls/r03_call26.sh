#!/bin/bash
# force loop without the "old value" copies (undef before the gather): parity subset, then K=100 A/B
set -o pipefail
OUT=gpurun_out/r03_call26; mkdir -p $OUT; export TMPDIR=/tmp
SPH_LIB_PATH=$PWD/cudafluidsimulator_amd/variants/libsph_hip_undef.so timeout -k 10 500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_properties.py tests/test_slab.py -m gpu -x -q > $OUT/pytest_undef.txt 2>&1; echo "pytest undef exit $?" | tee -a $OUT/summary.txt
tail -3 $OUT/pytest_undef.txt | tee -a $OUT/summary.txt
BENCH_ARGS="--steps 100 --warmup 5 --no-extra-legs" scripts/gpu_bench_variants.sh r03_call26/k100 main undef loose main undef | tee -a $OUT/summary.txt
