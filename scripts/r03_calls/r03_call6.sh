#!/bin/bash
set -o pipefail
OUT=gpurun_out/r03_call6; mkdir -p $OUT; export TMPDIR=/tmp
timeout -k 10 700 python -m pytest tests -m gpu -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/summary.txt
tail -6 $OUT/pytest.log | tee -a $OUT/summary.txt
timeout -k 10 120 python scripts/mgpu_loopback_study.py 4194304 20 random 2,4,8 $OUT/mgpu_4m.json 2>&1 | tee -a $OUT/summary.txt
timeout -k 10 200 python scripts/mgpu_loopback_study.py 16777216 10 random 4 $OUT/mgpu_16m.json 2>&1 | tee -a $OUT/summary.txt
timeout -k 10 300 python scripts/mgpu_loopback_study.py 67108864 3 grid 8 $OUT/mgpu_67m.json 2>&1 | tee -a $OUT/summary.txt
BENCH_ARGS="--steps 100 --warmup 5 --no-extra-legs" scripts/gpu_bench_variants.sh r03_call6/k100 main | tee -a $OUT/summary.txt
BENCH_ARGS="-n 262144 --steps 100 --warmup 5 --no-extra-legs" scripts/gpu_bench_variants.sh r03_call6/n262k main | tee -a $OUT/summary.txt
