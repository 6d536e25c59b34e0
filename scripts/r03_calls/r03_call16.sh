#!/bin/bash
set -o pipefail
OUT=gpurun_out/r03_call16; mkdir -p $OUT; export TMPDIR=/tmp
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py tests/test_cli.py tests/test_dropin.py -m gpu -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/summary.txt
tail -15 $OUT/pytest.log | tee -a $OUT/summary.txt
