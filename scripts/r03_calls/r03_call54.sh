#!/bin/bash
set -o pipefail
OUT=gpurun_out/r03_call54; mkdir -p $OUT; export TMPDIR=/tmp
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "graph or click" > $OUT/pytest1.txt 2>&1; echo "graph/click tests exit $? $(tail -1 $OUT/pytest1.txt)" | tee -a $OUT/summary.txt
SPH_GRAPH=1 timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_mgpu.py tests/test_slab.py -m gpu -x -q > $OUT/pytest2.txt 2>&1; echo "SPH_GRAPH=1 exit $? $(tail -1 $OUT/pytest2.txt)" | tee -a $OUT/summary.txt
