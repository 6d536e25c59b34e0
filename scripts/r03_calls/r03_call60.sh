#!/bin/bash
OUT=gpurun_out/r03_call60; mkdir -p $OUT; export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_sort.py -m gpu -q > $OUT/pytest.txt 2>&1; echo "sort tests exit $? $(tail -1 $OUT/pytest.txt)"; grep "^FAILED\|Error" $OUT/pytest.txt | head
