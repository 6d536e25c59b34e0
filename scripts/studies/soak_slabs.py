#!/usr/bin/env python3
"""Soak of the multi-GPU driver beyond the goldens: -n 4194304 -i random through step 300, 8 z-slabs (streams
transport: the RCCL path's stream layout, all slabs on this GPU; re-cut every 50 steps) against the single domain,
sha256 of the positions at steps 120/180/240/300 (GPU)."""
import hashlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import cudafluidsimulator_amd as sph
from cudafluidsimulator_amd import mgpu as M

n, CHECK = 4194304, (120, 180, 240, 300)
s = sph.default_settings(n, True)
want = {}
sim = sph.Simulator(s)
sim.setup()
for k in range(CHECK[-1]):
    sim.simulate()
    if k + 1 in CHECK:
        want[k + 1] = hashlib.sha256(sim.download_state()["pos"].tobytes()).hexdigest()[:16]
sim.close()
for world, transport, recut in ((8, "streams", 50), (5, "loopback", 0)):
    mg = M.MultiGpuSimulator(s, world=world, transport=transport, recut_every=recut)
    mg.setup()
    for k in range(CHECK[-1]):
        mg.simulate()
        if k + 1 in CHECK:
            got = hashlib.sha256(mg.download_state()["pos"].tobytes()).hexdigest()[:16]
            gp = hashlib.sha256(np.array(mg.getPosition()).tobytes()).hexdigest()[:16]
            print(world, transport, "step", k + 1, got, want[k + 1], "EQUAL" if got == want[k + 1] == gp else "DIFFER", flush=True)
    st = mg.stats()
    print(" owned per slab at the end:", list(st.owned[:world]))
    mg.close()
