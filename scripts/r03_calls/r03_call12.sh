#!/bin/bash
set -o pipefail
OUT=gpurun_out/r03_call12; mkdir -p $OUT; export TMPDIR=/tmp
timeout -k 10 700 python -m pytest tests -m gpu -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/summary.txt
tail -25 $OUT/pytest.log | tee -a $OUT/summary.txt
