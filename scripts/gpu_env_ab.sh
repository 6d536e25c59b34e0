#!/bin/bash
# A/B one environment knob of the library over several values in one gpurun call.
# usage: scripts/gpu_env_ab.sh TAG VAR value... (value "-" = unset); BENCH_ARGS extra bench flags
set -o pipefail
TAG=$1; VAR=$2; shift 2
OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
for v in "$@"; do
  if [ "$v" = "-" ]; then unset $VAR; else export $VAR=$v; fi
  timeout -k 10 200 python bench.py --cpu-steps 0 --no-fast-leg --no-linked-leg ${BENCH_ARGS:-} > $OUT/bench_$v.json 2> $OUT/bench_$v.err
  python - <<PY
import json
try:
    d=json.load(open("$OUT/bench_$v.json"))
    k=d["kernel_ms_per_step"]
    print("$VAR=$v: value %.4e ms/step %.3f | density %.3f force %.3f sort %.3f gather %.3f" % (d["value"], d["ms_per_step"], k["density"], k["force_integrate"], k["sort"], k["gather_cells"]), flush=True)
except Exception as e:
    print("$VAR=$v: bench failed", e); print(open("$OUT/bench_$v.err").read()[-1500:])
PY
done
