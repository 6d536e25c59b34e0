#!/bin/bash
set -o pipefail
OUT=gpurun_out/r03_call52; mkdir -p $OUT; export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/pytest.txt 2>&1; echo "pytest exit $?" | tee -a $OUT/summary.txt
tail -3 $OUT/pytest.txt | tee -a $OUT/summary.txt
BENCH_ARGS="--steps 100 --warmup 5" scripts/gpu_bench_variants.sh r03_call52/k100 main | tee -a $OUT/summary.txt
python - <<'PY' | tee -a gpurun_out/r03_call52/summary.txt
import json
d=json.load(open("gpurun_out/r03_call52/k100/bench_main.json"))
for k,v in d["other_configs"].items(): print(k, "%.4f ms/step"%v["ms_per_step"], {a:round(b,4) for a,b in v["kernel_ms_per_step"].items()})
PY
timeout -k 10 120 python scripts/mgpu_loopback_study.py 4194304 20 random 8 2>&1 | tail -1 | tee -a $OUT/summary.txt
