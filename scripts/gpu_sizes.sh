#!/bin/bash
# Bench the other BASELINE configurations on one GPU (kernel breakdown per config).
# usage: scripts/gpu_sizes.sh TAG
set -o pipefail
TAG=$1; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
for cfg in "8192 grid" "262144 random" "1295029 grid" "4194304 random"; do
  set -- $cfg
  timeout -k 10 200 python bench.py -n $1 --init $2 --cpu-steps 0 --no-fast-leg --no-linked-leg > $OUT/bench_$1.json 2> $OUT/bench_$1.err
  python - <<PY
import json
try:
    d=json.load(open("$OUT/bench_$1.json")); k=d["kernel_ms_per_step"]
    print("n=$1 $2: value %.4e ms/step %.4f | density %.4f force %.4f sort %.4f gather %.4f hash %.4f" % (d["value"], d["ms_per_step"], k["density"], k["force_integrate"], k["sort"], k["gather_cells"], k["hash"]), flush=True)
except Exception as e:
    print("n=$1: failed", e); print(open("$OUT/bench_$1.err").read()[-1500:])
PY
done
