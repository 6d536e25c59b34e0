#!/bin/bash
set -o pipefail
OUT=gpurun_out/r03_call5; mkdir -p $OUT; export TMPDIR=/tmp
timeout -k 10 700 python -m pytest tests -m gpu -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/summary.txt
tail -6 $OUT/pytest.log | tee -a $OUT/summary.txt
for v in main funi; do
  if [ "$v" = main ]; then unset SPH_LIB_PATH; else export SPH_LIB_PATH=$PWD/cudafluidsimulator_amd/variants/libsph_hip_$v.so; fi
  echo "== $v" | tee -a $OUT/summary.txt
  timeout -k 10 200 python scripts/studies/per_step_profile.py --every 10 > $OUT/profile_$v.txt 2>&1 && grep "^step" $OUT/profile_$v.txt | tee -a $OUT/summary.txt
done
unset SPH_LIB_PATH
BENCH_ARGS="--steps 100 --warmup 5 --no-extra-legs" scripts/gpu_bench_variants.sh r03_call5/k100 main funi | tee -a $OUT/summary.txt
BENCH_ARGS="--steps 20 --warmup 5 --no-extra-legs" scripts/gpu_bench_variants.sh r03_call5/k20 main funi | tee -a $OUT/summary.txt
BENCH_ARGS="-n 262144 --steps 100 --warmup 5 --no-extra-legs" scripts/gpu_bench_variants.sh r03_call5/n262k main | tee -a $OUT/summary.txt
