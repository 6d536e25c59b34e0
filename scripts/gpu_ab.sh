#!/bin/bash
# A/B several builds of libsph_hip in one gpurun call: parity subset + bench each.
# usage: scripts/gpu_ab.sh TAG [variant names...]   ("main" = the in-tree library)
set -o pipefail
TAG=$1; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
for v in "$@"; do
  if [ "$v" = main ]; then unset SPH_LIB_PATH; else export SPH_LIB_PATH=$PWD/cudafluidsimulator_amd/variants/libsph_hip_$v.so; fi
  echo "=== $v" | tee -a $OUT/status.txt
  timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -q -m gpu -p no:cacheprovider -x -k "steps_bit_exact or pressure or skewed or coincident or north_star" > $OUT/pytest_$v.log 2>&1
  echo "pytest $v exit $?" | tee -a $OUT/status.txt; tail -n 3 $OUT/pytest_$v.log
  timeout -k 10 200 python bench.py --cpu-steps 0 ${BENCH_ARGS:-} > $OUT/bench_$v.json 2> $OUT/bench_$v.err
  python - <<PY
import json
try:
    d=json.load(open("$OUT/bench_$v.json"))
    k=d["kernel_ms_per_step"]; r=d["roofline"]
    print("$v: value %.3e ms/step %.3f | density %.3f force %.3f sort %.3f gather %.3f hash %.3f | valu_frac d %.3f f %.3f" % (d["value"], d["ms_per_step"], k["density"], k["force_integrate"], k["sort"], k["gather_cells"], k["hash"], r.get("valu_frac",0), r.get("force_valu_frac",0)))
except Exception as e:
    print("$v: bench failed", e); print(open("$OUT/bench_$v.err").read()[-2000:])
PY
done
