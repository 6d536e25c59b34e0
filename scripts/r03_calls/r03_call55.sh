#!/bin/bash
# the whole GPU suite under the knobs that change the step's control flow
set -o pipefail
OUT=gpurun_out/r03_call55; mkdir -p $OUT; export TMPDIR=/tmp
for kv in SPH_GRAPH=1 SPH_PIPELINE=1 SPH_READBACK_SDMA=0 SPH_PIPELINE=0; do
  env $kv timeout -k 10 900 python -m pytest tests -m gpu -q > $OUT/pytest_$kv.txt 2>&1; echo "$kv exit $? $(tail -1 $OUT/pytest_$kv.txt)" | tee -a $OUT/summary.txt
  grep "^FAILED" $OUT/pytest_$kv.txt | tee -a $OUT/summary.txt
done
