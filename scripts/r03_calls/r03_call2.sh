#!/bin/bash
# round 3, GPU call 2: full parity suite; XCD rotation sweep; default bench line
set -o pipefail
OUT=gpurun_out/r03_call2; mkdir -p $OUT; export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests -m gpu -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/summary.txt
tail -8 $OUT/pytest.log | tee -a $OUT/summary.txt
for R in 0 1 2 4 8 16; do
  echo "== SPH_XCD_ROTATE=$R" | tee -a $OUT/summary.txt
  SPH_XCD_ROTATE=$R timeout -k 10 200 python scripts/studies/per_step_profile.py --every 10 > $OUT/profile_rot$R.txt 2>&1 && grep "^step" $OUT/profile_rot$R.txt | tee -a $OUT/summary.txt
done
( time timeout -k 10 500 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err ) 2>&1 | grep real | tee -a $OUT/summary.txt
python -c "
import json; d=json.load(open('$OUT/bench_default.json'))
print('default: ms/step %.3f value %.3e' % (d['ms_per_step'], d['value']), d['kernel_ms_per_step'])
print('cpu', d['cpu_baseline']['value'], d['cpu_baseline'].get('gpu_same_window'))
print('legs', d.get('other_configs'))
" | tee -a $OUT/summary.txt
