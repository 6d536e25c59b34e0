#!/bin/bash
set -o pipefail
OUT=gpurun_out/r03_call50; mkdir -p $OUT; export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/pytest.txt 2>&1; echo "pytest exit $?" | tee -a $OUT/summary.txt
tail -3 $OUT/pytest.txt | tee -a $OUT/summary.txt
timeout -k 10 120 python scripts/mgpu_loopback_study.py 4194304 20 random 2,4,8 $OUT/mgpu_4m.json 2>&1 | tee -a $OUT/summary.txt
TRANSPORT=streams timeout -k 10 120 python scripts/mgpu_loopback_study.py 4194304 20 random 8 2>&1 | tee -a $OUT/summary.txt
timeout -k 10 300 python scripts/mgpu_loopback_study.py 4194304 100 random 2,4,8 $OUT/mgpu_4m_100.json 2>&1 | tee -a $OUT/summary.txt
