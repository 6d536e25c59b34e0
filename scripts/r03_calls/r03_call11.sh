#!/bin/bash
set -o pipefail
OUT=gpurun_out/r03_call11; mkdir -p $OUT; export TMPDIR=/tmp
export SPH_LIB_PATH=$PWD/cudafluidsimulator_amd/variants/libsph_hip_dma96.so
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "headline or steps_bit_exact or zero_pair or dense" > $OUT/pytest_dma96.log 2>&1; echo "pytest(dma96) rc=$?" | tee -a $OUT/summary.txt
tail -3 $OUT/pytest_dma96.log | tee -a $OUT/summary.txt
unset SPH_LIB_PATH
BENCH_ARGS="--steps 100 --warmup 5 --no-extra-legs" scripts/gpu_bench_variants.sh r03_call11/k100 main dma160 dma96 dma64 dma32 main | tee -a $OUT/summary.txt
