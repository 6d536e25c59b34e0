#!/usr/bin/env python3
"""What the z-slab decomposition costs per slab: N slabs of the n-particle domain stepped
by the C++ multi-GPU driver's loopback transport on ONE GPU (no RCCL time in it).
usage: [TRANSPORT=streams] python scripts/mgpu_loopback_study.py [n] [steps] [init] [N,N,...] [json_out]
  init: random | grid (n > 109^3: the dense-lattice extension)
TRANSPORT=streams: the RCCL path's per-slab streams on one GPU -- the slabs' kernels then run
side by side, so per-slab event times include each other; compare the wall time per step.
The strong-scaling CEILING of the decomposition is single-domain kernel ms / slowest slab's kernel
ms: what N GPUs could reach if the exchanges and the host cost nothing."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cudafluidsimulator_amd as sph
from cudafluidsimulator_amd import mgpu as M

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4194304
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
init = sys.argv[3] if len(sys.argv) > 3 else "random"
worlds = [int(x) for x in sys.argv[4].split(",")] if len(sys.argv) > 4 else [2, 4, 8]
out_path = sys.argv[5] if len(sys.argv) > 5 else None
s = sph.default_settings(n, init == "random")
sim = sph.Simulator(s, flags=4)
sim.setup()
for _ in range(min(3, steps)):
    sim.simulate()
sim.setup()
sim.kernel_times(reset=True)
for _ in range(steps):
    sim.simulate()
kt = sim.kernel_times()
single = dict(grid=(kt.hash + kt.sort + kt.gather) / steps * 1e3, density=kt.density / steps * 1e3,
              force=kt.force / steps * 1e3)
single["kernels"] = single["grid"] + single["density"] + single["force"]
print("n=%d -i %s, first %d steps; single domain: grid %.3f density %.3f force %.3f = %.3f ms/step" % (
    n, init, steps, single["grid"], single["density"], single["force"], single["kernels"]), flush=True)
sim.close()
rows = []
for N in worlds:
    mg = M.MultiGpuSimulator(s, world=N, transport=os.environ.get("TRANSPORT", "loopback"))
    mg.setup()
    for _ in range(min(3, steps)):
        mg.simulate()
    mg.setup()
    mg.stats(reset=True)
    mg.sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        mg.simulate()
    mg.sync()
    wall = (time.perf_counter() - t0) / steps * 1e3
    st = mg.stats()
    k = max(range(N), key=lambda q: st.kernel_s[q])
    row = dict(slabs=N, grid=st.grid_s[k] / steps * 1e3, density=st.density_s[k] / steps * 1e3,
               force=st.force_s[k] / steps * 1e3, slowest_slab_kernels=st.kernel_s[k] / steps * 1e3,
               mean_slab_kernels=sum(st.kernel_s[:N]) / N / steps * 1e3, owned_slowest=int(st.owned[k]),
               wall_all_slabs_on_one_gpu=wall)
    row["strong_scaling_ceiling"] = single["kernels"] / row["slowest_slab_kernels"]
    rows.append(row)
    print("N=%d slowest slab: grid %.3f density %.3f force %.3f = %.3f ms/step (mean over slabs %.3f), owned %d; "
          "ceiling %.2fx; all %d slabs on this GPU: %.3f ms wall per step" % (
              N, row["grid"], row["density"], row["force"], row["slowest_slab_kernels"], row["mean_slab_kernels"],
              row["owned_slowest"], row["strong_scaling_ceiling"], N, wall), flush=True)
    mg.close()
if out_path:
    json.dump(dict(n=n, init=init, steps=steps, transport=os.environ.get("TRANSPORT", "loopback"),
                   single_domain=single, slabs=rows), open(out_path, "w"), indent=1)
