#!/bin/bash
set -o pipefail
OUT=gpurun_out/r03_call46; mkdir -p $OUT; export TMPDIR=/tmp
timeout -k 10 300 python scripts/mgpu_loopback_study.py 4194304 100 random 2,4,8 $OUT/mgpu_4m_100.json 2>&1 | tee -a $OUT/summary.txt
