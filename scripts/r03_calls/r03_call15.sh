#!/bin/bash
set -o pipefail
OUT=gpurun_out/r03_call15; mkdir -p $OUT; export TMPDIR=/tmp
timeout -k 10 700 python -m pytest tests -m gpu -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/summary.txt
tail -8 $OUT/pytest.log | tee -a $OUT/summary.txt
for P in 1 0; do
  export SPH_PIPELINE=$P; echo "== SPH_PIPELINE=$P" | tee -a $OUT/summary.txt
  BENCH_ARGS="--steps 20 --warmup 5 --no-extra-legs" scripts/gpu_bench_variants.sh r03_call15/k20_$P main | tee -a $OUT/summary.txt
  BENCH_ARGS="--steps 100 --warmup 5 --no-extra-legs" scripts/gpu_bench_variants.sh r03_call15/k100_$P main | tee -a $OUT/summary.txt
  BENCH_ARGS="-n 262144 --steps 100 --warmup 5 --no-extra-legs" scripts/gpu_bench_variants.sh r03_call15/n262k_$P main | tee -a $OUT/summary.txt
done
unset SPH_PIPELINE
SPH_STEP_TRACE=1 python bench.py --steps 20 --warmup 5 --cpu-steps 0 --no-linked-leg --no-fast-leg --no-count-replay --no-extra-legs 2>&1 >/dev/null | grep -A1 "step trace" | tee -a $OUT/summary.txt
