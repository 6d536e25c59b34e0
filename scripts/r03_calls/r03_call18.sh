#!/bin/bash
set -o pipefail
OUT=gpurun_out/r03_call18; mkdir -p $OUT; export TMPDIR=/tmp
timeout -k 10 700 python -m pytest tests -m gpu -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/summary.txt
tail -8 $OUT/pytest.log | tee -a $OUT/summary.txt
for S in 1 0; do
  export SPH_READBACK_SDMA=$S; echo "== SPH_READBACK_SDMA=$S" | tee -a $OUT/summary.txt
  BENCH_ARGS="--steps 20 --warmup 5 --no-extra-legs" scripts/gpu_bench_variants.sh r03_call18/k20_$S main | tee -a $OUT/summary.txt
  BENCH_ARGS="--steps 100 --warmup 5 --no-extra-legs" scripts/gpu_bench_variants.sh r03_call18/k100_$S main | tee -a $OUT/summary.txt
  BENCH_ARGS="-n 262144 --steps 100 --warmup 5 --no-extra-legs" scripts/gpu_bench_variants.sh r03_call18/n262k_$S main | tee -a $OUT/summary.txt
done
unset SPH_READBACK_SDMA
bash scripts/gpu_ktrace.sh r03_call18/ktrace 20 --warmup 5 | head -14 | tee -a $OUT/summary.txt
