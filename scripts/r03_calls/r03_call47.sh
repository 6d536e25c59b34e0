#!/bin/bash
# why a 1/8 slab of the headline run takes 3x its share of the single domain's force sweep (100 steps)
set -o pipefail
OUT=gpurun_out/r03_call47; mkdir -p $OUT; export TMPDIR=/tmp
run() { echo "== $1" | tee -a $OUT/summary.txt; timeout -k 10 300 python scripts/mgpu_loopback_study.py 4194304 100 random 8 2>&1 | grep "N=8\|single" | tee -a $OUT/summary.txt; }
run "main"
SPH_LIB_PATH=$PWD/cudafluidsimulator_amd/variants/libsph_hip_nodeal.so run "nodeal (64-row workgroups, rows in place)"
SPH_XCD_ROTATE=1 run "main, chunk->XCD map rotated every z-layer"
SPH_TILE_CHUNK=0 run "main, contiguous eighths"
SPH_TILE_CHUNK=1 run "main, chunks of one tile (round-robin)"
