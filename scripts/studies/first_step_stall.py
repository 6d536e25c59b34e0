#!/usr/bin/env python3
"""Which call of the first step after setup() holds the host?  (phase API, host timers)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import cudafluidsimulator_amd as sph

n = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
sim = sph.Simulator(sph.default_settings(n, True))
for rep in range(3):
    sim.setup()
    for _ in range(2):
        sim.simulate()
    sim.sync()
    t0 = time.perf_counter()
    sim.setup()
    t1 = time.perf_counter()
    sim.sync()
    if os.environ.get("SLEEP_MS"):
        time.sleep(float(os.environ["SLEEP_MS"]) * 1e-3)
    out = ["setup %.2f" % ((t1 - t0) * 1e3)]
    for step in range(3):
        for ph in ("grid", "density", "force", "readback"):
            a = time.perf_counter()
            sim.phase(ph)
            b = time.perf_counter()
            sim.sync()
            c = time.perf_counter()
            out.append("%s %.2f+%.2f" % (ph, (b - a) * 1e3, (c - b) * 1e3))
        out.append("|")
    print("rep %d (call + sync, ms): %s" % (rep, " ".join(out)))
sim.close()
