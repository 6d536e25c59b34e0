#!/bin/bash
set -o pipefail
OUT=gpurun_out/r03_call56; mkdir -p $OUT; export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_api_fuzz.py -m gpu -q > $OUT/pytest.txt 2>&1; echo "fuzz exit $? $(tail -1 $OUT/pytest.txt)" | tee -a $OUT/summary.txt
grep "^FAILED\|Error" $OUT/pytest.txt | head -20 | tee -a $OUT/summary.txt; true
