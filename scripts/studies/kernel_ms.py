#!/usr/bin/env python3
"""Kernel ms per step of the single domain over the first `steps` steps (untimed steps: no read-back).
  python scripts/studies/kernel_ms.py n steps [random|grid]     (SPH_LIB_PATH selects a variant library)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import cudafluidsimulator_amd as sph

n, steps = int(sys.argv[1]), int(sys.argv[2])
init = sys.argv[3] if len(sys.argv) > 3 else "random"
sim = sph.Simulator(sph.default_settings(n, init == "random"))
sim.setup()
for _ in range(min(3, steps)):
    sim.simulate()
sim.setup()
sim.kernel_times(reset=True)
for _ in range(steps):
    sim.simulate()
kt = sim.kernel_times()
print("%s n=%d -i %s steps 1..%d: grid %.3f density %.3f force %.3f ms/step" % (
    os.path.basename(os.environ.get("SPH_LIB_PATH", "main")), n, init, steps,
    (kt.hash + kt.sort + kt.gather) / steps * 1e3, kt.density / steps * 1e3, kt.force / steps * 1e3), flush=True)
sim.close()
