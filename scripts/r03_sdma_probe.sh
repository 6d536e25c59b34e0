env | grep -i "^HSA\|^ROC\|^GPU_\|^HIP" > gpurun_out/box_env.txt; cat gpurun_out/box_env.txt
export HSA_ENABLE_SDMA=1
bash scripts/gpu_ktrace.sh r03_ktrace_sdma1 20 --warmup 5 | head -8
python - <<'PY'
import json; d=json.load(open("gpurun_out/r03_ktrace_sdma1/prof_bench.json")); print("SDMA=1 under rocprof: ms/step", d["ms_per_step"], d["kernel_ms_per_step"])
PY
python bench.py --steps 20 --warmup 5 --cpu-steps 0 --no-linked-leg --no-fast-leg --no-count-replay --no-extra-legs > gpurun_out/bench_sdma1.json 2>/dev/null
python - <<'PY'
import json; d=json.load(open("gpurun_out/bench_sdma1.json")); print("SDMA=1: ms/step", d["ms_per_step"], d["kernel_ms_per_step"])
PY
export HSA_ENABLE_SDMA=0
python bench.py --steps 20 --warmup 5 --cpu-steps 0 --no-linked-leg --no-fast-leg --no-count-replay --no-extra-legs > gpurun_out/bench_sdma0.json 2>/dev/null
python - <<'PY'
import json; d=json.load(open("gpurun_out/bench_sdma0.json")); print("SDMA=0: ms/step", d["ms_per_step"], d["kernel_ms_per_step"])
PY
