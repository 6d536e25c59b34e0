#!/usr/bin/env python3
"""Soak beyond the goldens: -n 4194304 -i random through step 400, list vs lds sweeps, sha256 of the
positions at steps 150/200/260/330/400 (GPU; the two families must agree bit for bit).  The list run goes
through TIMED steps (simulateAndTime: read-back through the SDMA engine, getPosition() at every checkpoint),
the lds run through untimed ones."""
import sys, hashlib
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import cudafluidsimulator_amd as sph
n, steps = 4194304, 400
CHECK = (150, 200, 260, 330, 400)
out = {}
for sweep in ("list", "lds"):
    sim = sph.Simulator(sph.default_settings(n, True), sweep=sweep)
    sim.setup()
    t = sph.Times()
    for k in range(steps):
        if sweep == "list":
            sim.simulateAndTime(t)
        else:
            sim.simulate()
        if (k + 1) in CHECK:
            st = sim.download_state()
            out[(sweep, k + 1)] = hashlib.sha256(st["pos"].tobytes()).hexdigest()[:16]
            assert np.array_equal(np.array(sim.getPosition()).view(np.uint32), st["pos"].view(np.uint32)), "getPosition() != state"
    kt = sim.kernel_times()
    print(sweep, "density %.3f force %.3f ms/step avg" % (kt.density / steps * 1e3, kt.force / steps * 1e3), flush=True)
    g = sim.download_grid()
    cnt = g["cells"][:, 1] - g["cells"][:, 0]
    print("max particles per cell", cnt.max(), "rho max", float(st["rho"].max()))
    sim.close()
for s in CHECK:
    print(s, out[("list", s)], out[("lds", s)], "EQUAL" if out[("list", s)] == out[("lds", s)] else "DIFFER")
