#!/usr/bin/env python3
"""What the z-slab decomposition costs per slab: N slabs of the n-particle domain stepped
by the C++ multi-GPU driver's loopback transport on ONE GPU (no RCCL time in it).
usage: [TRANSPORT=streams] python scripts/mgpu_loopback_study.py [n] [steps]
TRANSPORT=streams: the RCCL path's per-slab streams on one GPU -- the slabs' kernels then run
side by side, so per-slab event times include each other; compare the wall time per step."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cudafluidsimulator_amd as sph
from cudafluidsimulator_amd import mgpu as M

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4194304
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
s = sph.default_settings(n, True)
sim = sph.Simulator(s, flags=4)
sim.setup()
for _ in range(3):
    sim.simulate()
sim.setup()
sim.kernel_times(reset=True)
for _ in range(steps):
    sim.simulate()
kt = sim.kernel_times()
print("single domain: grid %.3f density %.3f force %.3f ms/step" % (
    (kt.hash + kt.sort + kt.gather) / steps * 1e3, kt.density / steps * 1e3, kt.force / steps * 1e3))
sim.close()
for N in (2, 4, 8):
    mg = M.MultiGpuSimulator(s, world=N, transport=os.environ.get("TRANSPORT", "loopback"))
    mg.setup()
    for _ in range(3):
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
    print("N=%d slowest slab: grid %.3f density %.3f force %.3f = %.3f ms/step (mean over slabs %.3f), owned %d" % (
        N, st.grid_s[k] / steps * 1e3, st.density_s[k] / steps * 1e3, st.force_s[k] / steps * 1e3,
        st.kernel_s[k] / steps * 1e3, sum(st.kernel_s[:N]) / N / steps * 1e3, st.owned[k]) + "; all %d slabs on this GPU: %.3f ms wall per step" % (N, wall))
    mg.close()
