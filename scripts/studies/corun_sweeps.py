#!/usr/bin/env python3
"""Do the density sweep (VALU-bound) and the force sweep (texture / L1-bound) run better SIDE BY SIDE than one
after the other?  Two independent simulators of the headline configuration on one GPU (own streams, no read-back),
stepped 100 times (a) one whole step after the other, synchronised; (b) phase-shifted through the phase API so
that A's force sweep is queued beside B's grid build + density sweep and vice versa.  Wall time of the late steps."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import cudafluidsimulator_amd as sph
from cudafluidsimulator_amd import _lib

n, steps = 4194304, 100
FIRST = 60  # time steps FIRST..steps (the dense floor)


def make():
    s = sph.Simulator(sph.default_settings(n, True), flags=_lib.SPH_FLAG_NO_READBACK)
    s.setup()
    return s


def full_step(s):
    for ph in ("grid", "density", "force", "readback"):
        s.phase(ph)


# (a) sequential: every step of A, then of B, each synchronised
A, B = make(), make()
for k in range(steps):
    if k == FIRST:
        A.sync(); B.sync(); t0 = time.perf_counter()
    full_step(A); A.sync()
    full_step(B); B.sync()
A.sync(); B.sync()
seq = time.perf_counter() - t0
A.close(); B.close()

# (b) phase-shifted: A is half a step ahead
A, B = make(), make()
A.phase("grid"); A.phase("density")
for k in range(steps):
    if k == FIRST:
        A.sync(); B.sync(); t0 = time.perf_counter()
    A.phase("force"); A.phase("readback")          # A's force sweep ...
    B.phase("grid"); B.phase("density")            # ... beside B's grid build and density sweep
    if k + 1 < steps:
        B.phase("force"); B.phase("readback")
        A.phase("grid"); A.phase("density")
    else:
        B.phase("force"); B.phase("readback")
    if k % 4 == 3:  # keep the two queues from running away from each other
        A.sync(); B.sync()
A.sync(); B.sync()
par = time.perf_counter() - t0
A.close(); B.close()
print("two simulators, steps %d..%d: one after the other %.2f ms per step pair, phase-shifted side by side %.2f (%.1f %%)" % (
    FIRST + 1, steps, seq / (steps - FIRST) * 1e3, par / (steps - FIRST) * 1e3, 100 * (par / seq - 1)))
