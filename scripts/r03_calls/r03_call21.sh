#!/bin/bash
set -o pipefail
OUT=gpurun_out/r03_call21; mkdir -p $OUT; export TMPDIR=/tmp
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "mixed_read_back or grid_built_ahead" > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/summary.txt
tail -6 $OUT/pytest.log | tee -a $OUT/summary.txt
python bench.py --mode display --steps 30 --warmup 3 --no-extra-legs --cpu-steps 0 --no-linked-leg --no-fast-leg --no-count-replay > $OUT/display.json 2>$OUT/display.err; python -c "
import json; d=json.load(open('$OUT/display.json')); print('display mode ms/frame', d['ms_per_step'], d['kernel_ms_per_step'])" | tee -a $OUT/summary.txt
python bench.py --mode free --steps 30 --warmup 3 --no-extra-legs --cpu-steps 0 --no-linked-leg --no-fast-leg --no-count-replay > $OUT/free.json 2>$OUT/free.err; python -c "
import json; d=json.load(open('$OUT/free.json')); print('free mode ms/step', d['ms_per_step'])" | tee -a $OUT/summary.txt
