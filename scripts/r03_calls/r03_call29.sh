#!/bin/bash
# adjacent-row LDS windows in the force sweep: parity subset under the variant, then K=100 A/B
set -o pipefail
OUT=gpurun_out/r03_call29; mkdir -p $OUT; export TMPDIR=/tmp
SPH_LIB_PATH=$PWD/cudafluidsimulator_amd/variants/libsph_hip_adj352.so timeout -k 10 500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_properties.py tests/test_slab.py -m gpu -x -q > $OUT/pytest_adj.txt 2>&1; echo "pytest adj352 exit $?" | tee -a $OUT/summary.txt
tail -3 $OUT/pytest_adj.txt | tee -a $OUT/summary.txt
BENCH_ARGS="--steps 100 --warmup 5 --no-extra-legs" scripts/gpu_bench_variants.sh r03_call29/k100 adj0 adj352 adj304 adj0 adj352 | tee -a $OUT/summary.txt
