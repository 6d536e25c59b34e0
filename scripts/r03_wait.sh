run() { python bench.py --steps 20 --warmup 5 --cpu-steps 0 --no-linked-leg --no-fast-leg --no-count-replay --no-extra-legs > gpurun_out/bench_wait.json 2>/dev/null
  python - <<PY
import json; d=json.load(open("gpurun_out/bench_wait.json")); k=d["kernel_ms_per_step"]; print("$1: ms/step %.3f kernels %.3f readback %.3f" % (d["ms_per_step"], sum(v for a,v in k.items() if a!="readback_d2h"), k["readback_d2h"]))
PY
}
run default
ROC_ACTIVE_WAIT_TIMEOUT=5000 run active_wait_5ms
HSA_ENABLE_INTERRUPT=0 run no_interrupt
ROC_ACTIVE_WAIT_TIMEOUT=5000 HSA_ENABLE_INTERRUPT=0 run both
run default
