#!/bin/bash
# the parity tests under every library knob flipped (one at a time)
set -o pipefail
OUT=gpurun_out/r03_call53; mkdir -p $OUT; export TMPDIR=/tmp
for kv in SPH_ZERO_PAIR_FILTER=0 SPH_PIPELINE=1 SPH_PIPELINE=0 SPH_READBACK_SDMA=0 SPH_GRAPH=1 SPH_SLIM_DIV=0 SPH_XCD_ROTATE=3 SPH_TILE_CHUNK=1 SPH_MGPU_THREADS=1; do
  env $kv timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_mgpu.py tests/test_slab.py -m gpu -x -q > $OUT/pytest_$kv.txt 2>&1; echo "$kv exit $? $(tail -1 $OUT/pytest_$kv.txt)" | tee -a $OUT/summary.txt
done
