#!/bin/bash
# Bench several builds of libsph_hip in one gpurun call (no parity run: for perf-only
# experiment builds whose results are wrong by construction, and quick A/Bs).
# usage: scripts/gpu_bench_variants.sh TAG [variant names...]   ("main" = the in-tree library)
set -o pipefail
TAG=$1; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
for v in "$@"; do
  if [ "$v" = main ]; then unset SPH_LIB_PATH; else export SPH_LIB_PATH=$PWD/cudafluidsimulator_amd/variants/libsph_hip_$v.so; fi
  timeout -k 10 200 python bench.py --cpu-steps 0 --no-linked-leg --no-fast-leg ${BENCH_ARGS:---steps 20 --warmup 3} > $OUT/bench_$v.json 2> $OUT/bench_$v.err
  python - <<PY
import json
try:
    d=json.load(open("$OUT/bench_$v.json"))
    k=d["kernel_ms_per_step"]
    print("$v: ms/step %.3f | density %.3f force %.3f sort %.3f gather %.3f hash %.3f" % (d["ms_per_step"], k["density"], k["force_integrate"], k["sort"], k["gather_cells"], k["hash"]))
except Exception as e:
    print("$v: bench failed", e); print(open("$OUT/bench_$v.err").read()[-1500:])
PY
done
