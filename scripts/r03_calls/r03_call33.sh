#!/bin/bash
# slab studies on the sorted-lanes build (first 20 steps and a late window)
set -o pipefail
OUT=gpurun_out/r03_call33; mkdir -p $OUT; export TMPDIR=/tmp
timeout -k 10 120 python scripts/mgpu_loopback_study.py 4194304 20 random 2,4,8 $OUT/mgpu_4m.json 2>&1 | tee -a $OUT/summary.txt
timeout -k 10 200 python scripts/mgpu_loopback_study.py 16777216 10 random 4 $OUT/mgpu_16m.json 2>&1 | tee -a $OUT/summary.txt
timeout -k 10 300 python scripts/mgpu_loopback_study.py 67108864 3 grid 8 $OUT/mgpu_67m.json 2>&1 | tee -a $OUT/summary.txt
timeout -k 10 300 python bench.py --loopback-slabs 8 --steps 100 --warmup 5 --cpu-steps 0 --no-extra-legs > $OUT/bench_loopback8_k100.json 2> $OUT/bench_loopback8.err; echo "bench loopback8 exit $?" | tee -a $OUT/summary.txt
python -c "
import json; d=json.load(open('$OUT/bench_loopback8_k100.json')); print(d['ms_per_step'], d['value'], d.get('mgpu'))" | tee -a $OUT/summary.txt
