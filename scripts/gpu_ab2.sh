#!/bin/bash
# Fast A/B: tiny parity subset + bench per variant.  usage: gpu_ab2.sh TAG variants...
set -o pipefail
TAG=$1; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
for v in "$@"; do
  if [ "$v" = main ]; then unset SPH_LIB_PATH; else export SPH_LIB_PATH=$PWD/cudafluidsimulator_amd/variants/libsph_hip_$v.so; fi
  timeout -k 10 120 python -m pytest tests/test_gpu_parity.py -q -m gpu -p no:cacheprovider -x -k "pressure or skewed or coincident" > $OUT/pytest_$v.log 2>&1
  echo "pytest $v exit $? : $(tail -n 1 $OUT/pytest_$v.log)"
  for rep in 1 2; do
  timeout -k 10 200 python bench.py --cpu-steps 0 ${BENCH_ARGS:-} > $OUT/bench_${v}_$rep.json 2> $OUT/bench_${v}_$rep.err
  python - <<PY
import json
try:
    d=json.load(open("$OUT/bench_${v}_$rep.json")); k=d["kernel_ms_per_step"]; r=d["roofline"]
    print("$v#$rep: value %.3e ms/step %.3f | density %.3f force %.3f sort %.3f gather %.3f hash %.3f | valu_frac d %.3f f %.3f" % (d["value"], d["ms_per_step"], k["density"], k["force_integrate"], k["sort"], k["gather_cells"], k["hash"], r.get("valu_frac",0), r.get("force_valu_frac",0)))
except Exception as e:
    print("$v: bench failed", e); print(open("$OUT/bench_${v}_$rep.err").read()[-1500:])
PY
  done
done
