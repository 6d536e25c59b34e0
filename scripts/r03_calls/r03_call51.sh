#!/bin/bash
# density sweep: a select-free, vote-free first phase per run (fixed trip count)
set -o pipefail
OUT=gpurun_out/r03_call51; mkdir -p $OUT; export TMPDIR=/tmp
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_properties.py tests/test_slab.py -m gpu -x -q > $OUT/pytest.txt 2>&1; echo "pytest exit $?" | tee -a $OUT/summary.txt
tail -2 $OUT/pytest.txt | tee -a $OUT/summary.txt
BENCH_ARGS="--steps 100 --warmup 5 --no-extra-legs" scripts/gpu_bench_variants.sh r03_call51/k100 main p1w5 nop1 main p1w5 nop1 | tee -a $OUT/summary.txt
