#!/usr/bin/env python3
"""Where does a step's wall time go without the per-step synchronisation of -m time?
usage: python scripts/studies/free_mode_overheads.py [n] [steps]"""
import os
if os.environ.get("WITH_TORCH"):
    import torch  # noqa: F401  (its bundled HIP runtime gets loaded first)
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import cudafluidsimulator_amd as sph
from cudafluidsimulator_amd import _lib

n = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
K = int(sys.argv[2]) if len(sys.argv) > 2 else 100
s = sph.default_settings(n, True)


def run(label, flags=0, timed=False, sync_every=0):
    sim = sph.Simulator(s, flags=flags)
    sim.setup()
    t = sph.Times()
    for _ in range(5):
        sim.simulate()
    sim.sync()
    sim.setup()
    sim.kernel_times(reset=True)
    t0 = time.perf_counter()
    host = 0.0
    for k in range(K):
        h0 = time.perf_counter()
        if timed:
            sim.simulateAndTime(t)
        else:
            sim.simulate()
        host += time.perf_counter() - h0
        if sync_every and (k + 1) % sync_every == 0:
            sim.sync()
    sim.sync()
    el = time.perf_counter() - t0
    kt = sim.kernel_times()
    ksum = (kt.hash + kt.sort + kt.gather + kt.density + kt.force) / K * 1e3
    print("%-44s %.3f ms/step wall | host time in the step call %.3f | kernels %.3f | readback %.3f" % (
        label, el / K * 1e3, host / K * 1e3, ksum, kt.readback / K * 1e3))
    sim.close()


# clocks up first: a second of stepping
_w = sph.Simulator(sph.default_settings(4194304, True))
_w.setup()
for _ in range(400):
    _w.simulate()
_w.sync()
_w.close()
run("simulateAndTime (sync every step)", timed=True)
run("simulate, no sync")
run("simulate, sync every step", sync_every=1)
run("simulate, sync every 4 steps", sync_every=4)
run("simulate, no read-back, no sync", flags=_lib.SPH_FLAG_NO_READBACK)
run("simulate, no read-back, sync every step", flags=_lib.SPH_FLAG_NO_READBACK, sync_every=1)
run("simulateAndTime (sync every step) again", timed=True)
run("simulate, no sync, again")
