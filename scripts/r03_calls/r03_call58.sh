#!/bin/bash
OUT=gpurun_out/r03_call58; mkdir -p $OUT; export TMPDIR=/tmp
RT=$(find /opt/rocm/lib/llvm/lib/clang -name "libclang_rt.ubsan_standalone-x86_64.so" | head -1)
export LD_LIBRARY_PATH=$(dirname $RT):$LD_LIBRARY_PATH
export UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=0
export SPH_LIB_PATH=$PWD/cudafluidsimulator_amd/variants/libsph_hip_hostubsan.so
export SPH_MGPU_LIB_PATH=$PWD/cudafluidsimulator_amd/variants/libsph_mgpu_hostubsan.so
timeout -k 10 600 python -X faulthandler -m pytest -s tests/test_gpu_api_fuzz.py tests/test_gpu_parity.py tests/test_mgpu.py tests/test_slab.py -m gpu -q > $OUT/pytest.txt 2> $OUT/stderr.txt; echo "exit $?"
tail -3 $OUT/pytest.txt; grep -c "runtime error" $OUT/stderr.txt; grep "runtime error" $OUT/stderr.txt | sort | uniq -c | sort -rn | head -20; tail -20 $OUT/stderr.txt
