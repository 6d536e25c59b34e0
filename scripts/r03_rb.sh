run() { python bench.py --steps 20 --warmup 5 --cpu-steps 0 --no-linked-leg --no-fast-leg --no-count-replay --no-extra-legs > gpurun_out/bench_rb.json 2>/dev/null
  python - <<PY
import json; d=json.load(open("gpurun_out/bench_rb.json")); k=d["kernel_ms_per_step"]; print("$1: ms/step %.3f sort %.3f gather %.3f density %.3f force %.3f readback %.3f exposed %.4f" % (d["ms_per_step"], k["sort"], k["gather_cells"], k["density"], k["force_integrate"], k["readback_d2h"], d["m_time_table_s"]["data_transfer_exposed"]))
PY
}
run default
SPH_HOSTPOS_NONCOHERENT=1 run noncoherent
SPH_COPY_CHUNKS=4 run chunks4
SPH_COPY_CHUNKS=16 run chunks16
SPH_COPY_CHUNKS=64 run chunks64
SPH_HOSTPOS_NONCOHERENT=1 SPH_COPY_CHUNKS=16 run noncoherent_chunks16
