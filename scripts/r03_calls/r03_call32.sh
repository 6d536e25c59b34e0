#!/bin/bash
set -o pipefail
OUT=gpurun_out/r03_call32; mkdir -p $OUT; export TMPDIR=/tmp
t0=$(date +%s.%N)
timeout -k 10 600 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench exit $?" | tee -a $OUT/summary.txt
t1=$(date +%s.%N); echo "wall $(echo "$t1 - $t0" | bc) s" | tee -a $OUT/summary.txt
python - <<'PY' | tee -a gpurun_out/r03_call32/summary.txt
import json
d=json.load(open("gpurun_out/r03_call32/bench_default.json"))
print({k:d[k] for k in ("metric","value","unit","n_gpus","steps","warmup","ms_per_step","vs_baseline","dtype")})
print(d["roofline"]); print(d["cpu_baseline"]); print(d.get("full_run_100"))
PY
