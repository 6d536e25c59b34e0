#!/bin/bash
set -o pipefail
OUT=gpurun_out/r03_call30; mkdir -p $OUT; export TMPDIR=/tmp
SPH_LIB_PATH=$PWD/cudafluidsimulator_amd/variants/libsph_hip_depth3.so timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > $OUT/pytest_d3.txt 2>&1; echo "pytest depth3 exit $?" | tee -a $OUT/summary.txt
tail -2 $OUT/pytest_d3.txt | tee -a $OUT/summary.txt
BENCH_ARGS="--steps 100 --warmup 5 --no-extra-legs" scripts/gpu_bench_variants.sh r03_call30/k100 adj0 depth3 adj0 depth3 | tee -a $OUT/summary.txt
