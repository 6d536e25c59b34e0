#!/usr/bin/env python3
"""Per-step wall time of simulateAndTime (one host sync per step) against the step's own
kernel time: is the difference a constant gap or a few long stalls?
usage: [WITH_TORCH=1] python scripts/studies/step_jitter.py [n] [steps]"""
import os
if os.environ.get("WITH_TORCH"):
    import torch  # noqa: F401  (its bundled HIP runtime gets loaded first)
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import cudafluidsimulator_amd as sph

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4194304
K = int(sys.argv[2]) if len(sys.argv) > 2 else 100
sim = sph.Simulator(sph.default_settings(n, True))
for rep in range(3):
    sim.setup()
    t = sph.Times()
    for _ in range(5):
        sim.simulateAndTime(t)
    sim.sync()
    sim.setup()
    t = sph.Times()
    w = np.zeros(K)
    k = np.zeros(K)
    for i in range(K):
        b = t.buildGrid + t.sphUpdate
        t0 = time.perf_counter()
        sim.simulateAndTime(t)
        w[i] = time.perf_counter() - t0
        k[i] = t.buildGrid + t.sphUpdate - b
    g = (w - k) * 1e3
    print("rep %d: wall %.3f ms/step, kernels %.3f, gap mean %.3f median %.3f p90 %.3f max %.3f (at step %d); steps with gap > 0.2 ms: %d" % (
        rep, w.mean() * 1e3, k.mean() * 1e3, g.mean(), np.median(g), np.percentile(g, 90), g.max(), g.argmax() + 1, (g > 0.2).sum()))
sim.close()
