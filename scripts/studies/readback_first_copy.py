#!/usr/bin/env python3
"""Per-step duration of the position read-back (copy-stream events) and wall time right after
setup(): which copy is slow?   usage: python scripts/studies/readback_first_copy.py [n] [steps]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import cudafluidsimulator_amd as sph

n = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
K = int(sys.argv[2]) if len(sys.argv) > 2 else 8
sim = sph.Simulator(sph.default_settings(n, True))
for rep in range(3):
    sim.setup()
    t = sph.Times()
    for _ in range(2):
        sim.simulateAndTime(t)
    sim.sync()
    sim.setup()
    sim.sync()
    line = []
    for i in range(K):
        sim.kernel_times(reset=True)
        t0 = time.perf_counter()
        sim.simulateAndTime(t)
        sim.sync()          # compute AND copy stream: this step's read-back is done
        w = time.perf_counter() - t0
        kt = sim.kernel_times()
        line.append("%.2f/%.2f" % (w * 1e3, kt.readback * 1e3))
    print("rep %d wall/readback ms per step: %s" % (rep, " ".join(line)))
sim.close()
