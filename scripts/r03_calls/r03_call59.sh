#!/bin/bash
# the CLI's host code (main.cpp, simulator.cpp, headless.cpp; g++) under AddressSanitizer + UBSan
OUT=gpurun_out/r03_call59; mkdir -p $OUT; export TMPDIR=/tmp
S=cudafluidsimulator_amd/variants/sph_asan
export ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1
run() { echo "== $*" | tee -a $OUT/summary.txt; env "${ENVV[@]}" timeout -k 10 200 $S "$@" > $OUT/out.txt 2> $OUT/err.txt; echo "exit $?" | tee -a $OUT/summary.txt; tail -3 $OUT/out.txt | tee -a $OUT/summary.txt; grep -i "sanitizer\|runtime error" $OUT/err.txt | head -5 | tee -a $OUT/summary.txt; }
ENVV=(SPH_PRINT_SHA256=1); run -n 20000 -i random -m time
ENVV=(SPH_PRINT_SHA256=1); run -n 8192 -i grid -m time
ENVV=(SPH_FREE_FRAMES=6 SPH_FREE_CLICK=1 SPH_PRINT_SHA256=1); run -n 4096 -i random -m free
ENVV=(SPH_PRINT_SHA256=1 SPH_GPUS=4 SPH_TRANSPORT=loopback); run -n 60000 -i random -m time
ENVV=(SPH_FREE_FRAMES=6 SPH_FREE_CLICK=1 SPH_PRINT_SHA256=1 SPH_GPUS=3 SPH_TRANSPORT=streams); run -n 30000 -i random -m free
ENVV=(X=1); run -n 100 -i bogus -m time
ENVV=(X=1); run -n 0 -i grid -m time
