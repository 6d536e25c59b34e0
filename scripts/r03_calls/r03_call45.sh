#!/bin/bash
set -o pipefail
OUT=gpurun_out/r03_call45; mkdir -p $OUT; export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_properties.py -m gpu -x -q -k "eight_slabs or config4" --durations=5 > $OUT/pytest.txt 2>&1; echo "pytest exit $?" | tee -a $OUT/summary.txt
tail -12 $OUT/pytest.txt | tee -a $OUT/summary.txt
