#!/bin/bash
OUT=gpurun_out/r03_call62; mkdir -p $OUT; export TMPDIR=/tmp
for st in 70 99; do
timeout -k 10 200 python scripts/studies/step_from_snapshot.py make $st /tmp/snap$st.sph 2>&1 | tail -1
for v in main noemit onereadnoemit; do
  if [ "$v" = main ]; then unset SPH_LIB_PATH; else export SPH_LIB_PATH=$PWD/cudafluidsimulator_amd/variants/libsph_hip_$v.so; fi
  timeout -k 10 120 python scripts/studies/step_from_snapshot.py time /tmp/snap$st.sph 3 2>&1 | tail -1 | tee -a $OUT/summary.txt
done; unset SPH_LIB_PATH; done
