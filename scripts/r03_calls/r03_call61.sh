#!/bin/bash
OUT=gpurun_out/r03_call61; mkdir -p $OUT; export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_api_fuzz.py -m gpu -q -k nasty > $OUT/pytest.txt 2>&1; echo "nasty exit $? $(tail -1 $OUT/pytest.txt)"; grep "AssertionError" $OUT/pytest.txt | cut -c1-400 | head
