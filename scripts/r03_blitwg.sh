for W in default 2 4 8 16 32 64; do
  if [ "$W" = default ]; then unset DEBUG_CLR_LIMIT_BLIT_WG; else export DEBUG_CLR_LIMIT_BLIT_WG=$W; fi
  python bench.py --steps 20 --warmup 5 --cpu-steps 0 --no-linked-leg --no-fast-leg --no-count-replay --no-extra-legs > gpurun_out/bench_blitwg_$W.json 2>/dev/null
  python - <<PY
import json; d=json.load(open("gpurun_out/bench_blitwg_$W.json")); k=d["kernel_ms_per_step"]; print("BLIT_WG=$W: ms/step %.3f sort %.3f gather %.3f density %.3f force %.3f readback %.3f exposed %.4f" % (d["ms_per_step"], k["sort"], k["gather_cells"], k["density"], k["force_integrate"], k["readback_d2h"], d["m_time_table_s"]["data_transfer_exposed"]))
PY
done
