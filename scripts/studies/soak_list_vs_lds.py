#!/usr/bin/env python3
"""Soak beyond the goldens: -n 4194304 -i random through step 260, list vs lds sweeps, sha256 of the
positions at steps 150/200/260 (GPU; the two families must agree bit for bit)."""
import sys, hashlib
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import cudafluidsimulator_amd as sph
n, steps = 4194304, 260
out = {}
for sweep in ("list", "lds"):
    sim = sph.Simulator(sph.default_settings(n, True), sweep=sweep)
    sim.setup()
    for k in range(steps):
        sim.simulate()
        if (k + 1) in (150, 200, 260):
            st = sim.download_state()
            out[(sweep, k + 1)] = hashlib.sha256(st["pos"].tobytes()).hexdigest()[:16]
    kt = sim.kernel_times()
    print(sweep, "density %.3f force %.3f ms/step avg" % (kt.density / steps * 1e3, kt.force / steps * 1e3), flush=True)
    g = sim.download_grid()
    cnt = g["cells"][:, 1] - g["cells"][:, 0]
    print("max particles per cell", cnt.max(), "rho max", float(st["rho"].max()))
    sim.close()
for s in (150, 200, 260):
    print(s, out[("list", s)], out[("lds", s)], "EQUAL" if out[("list", s)] == out[("lds", s)] else "DIFFER")
