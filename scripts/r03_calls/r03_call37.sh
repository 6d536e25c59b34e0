#!/bin/bash
set -o pipefail
OUT=gpurun_out/r03_call37; mkdir -p $OUT
for v in nosort ident main; do bash scripts/gpu_pmc_study.sh r03_call37 $v 2>&1 | tee -a $OUT/summary.txt; done
