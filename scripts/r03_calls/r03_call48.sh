#!/bin/bash
set -o pipefail
OUT=gpurun_out/r03_call48; mkdir -p $OUT; export TMPDIR=/tmp
run() { timeout -k 10 200 python bench.py --cpu-steps 0 --no-linked-leg --no-fast-leg --steps 100 --warmup 5 --no-extra-legs > $OUT/b.json 2> $OUT/b.err; python -c "
import json; d=json.load(open('$OUT/b.json')); k=d['kernel_ms_per_step']; print('$1: ms/step %.3f | density %.3f force %.3f sort %.3f gather %.3f' % (d['ms_per_step'], k['density'], k['force_integrate'], k['sort'], k['gather_cells']))" | tee -a $OUT/summary.txt; }
run "default"
SPH_XCD_ROTATE=1 run "rotate 1"
SPH_TILE_CHUNK=0 run "contiguous eighths"
SPH_TILE_CHUNK=1 run "round-robin tiles"
SPH_TILE_CHUNK=8 run "chunks of 8 tiles"
run "default"
