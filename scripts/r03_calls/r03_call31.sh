#!/bin/bash
# the round-end sequence the driver runs: GPU tests, smoke, default bench
set -o pipefail
OUT=gpurun_out/r03_call31; mkdir -p $OUT; export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/pytest.txt 2>&1; echo "pytest exit $?" | tee -a $OUT/summary.txt
tail -3 $OUT/pytest.txt | tee -a $OUT/summary.txt
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3 | tee -a $OUT/summary.txt
timeout -k 10 600 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench exit $?" | tee -a $OUT/summary.txt

python - <<'PY' | tee -a gpurun_out/r03_call31/summary.txt
import json
d=json.load(open("gpurun_out/r03_call31/bench_default.json"))
print({k:d[k] for k in ("metric","value","unit","n_gpus","steps","warmup","ms_per_step","vs_baseline","dtype")})
print(d["roofline"]); print(d["cpu_baseline"]); print(d.get("full_run_100"))
PY
