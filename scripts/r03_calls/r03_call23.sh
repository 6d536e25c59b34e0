#!/bin/bash
set -o pipefail
OUT=gpurun_out/r03_call23; mkdir -p $OUT; export TMPDIR=/tmp
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py tests/test_cli.py -m gpu -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/summary.txt
tail -3 $OUT/pytest.log | tee -a $OUT/summary.txt
SPH_STEP_TRACE=1 python bench.py --steps 20 --warmup 5 > $OUT/bench_default.json 2> $OUT/bench_default.err; grep "SDMA engine" $OUT/bench_default.err | sort | uniq -c | tee -a $OUT/summary.txt
python -c "
import json; d=json.load(open('$OUT/bench_default.json')); print(d['value'], d['ms_per_step'], d['kernel_ms_per_step']['readback_d2h']); print({k:(round(v['ms_per_step'],3), round(v['kernel_ms_per_step']['readback_d2h'],3)) for k,v in d['other_configs'].items()}); print(d['full_run_100']['ms_per_step'], d['full_run_100']['kernel_ms_per_step']['readback_d2h'])" | tee -a $OUT/summary.txt
