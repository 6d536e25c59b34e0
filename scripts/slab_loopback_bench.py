#!/usr/bin/env python3
"""How much work does the z-slab decomposition add?  Runs N slabs of one
n-particle domain inside ONE process on ONE GPU (loopback transport) and
compares the wall time per step with the single-domain path.  Per-rank time on a
real N-GPU node ~ (loopback total) / N + exchange latency."""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import cudafluidsimulator_amd as sph
from cudafluidsimulator_amd import slab as S

ap = argparse.ArgumentParser()
ap.add_argument("-n", type=int, default=4194304)
ap.add_argument("--slabs", type=int, nargs="+", default=[1, 2, 4, 8])
ap.add_argument("--steps", type=int, default=20)
a = ap.parse_args()
settings = sph.default_settings(a.n, True)
p4, v4 = S.make_initial(settings)
for world in a.slabs:
    bounds, parts = S.split_initial(p4, v4, settings.h, 100, world)
    slabs = []
    for r, ((zlo, zhi), (pp, vv)) in enumerate(zip(bounds, parts)):
        cap = int(len(pp) * 1.5) + 65536
        sl = S.Slab(S.HipSlabBackend(settings, cap, device=0), r, world, zlo, zhi, 100,
                    face_cap=min(S.default_face_cap(p4, settings.h, 100), cap))
        sl.load(torch.from_numpy(pp).cuda(), torch.from_numpy(vv).cuda())
        slabs.append(sl)
    S.run_loopback(slabs, 2)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    S.run_loopback(slabs, a.steps)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.steps
    kts = [sl.b.kernel_times() for sl in slabs]
    st = max(1, kts[0].steps)
    print(f"slabs {world}: {dt*1e3:.3f} ms/step total ({dt*1e3/world:.3f} per slab) | per-slab GPU ms: "
          f"sort {np.mean([k.sort for k in kts])/st*1e3:.3f} density {np.mean([k.density for k in kts])/st*1e3:.3f} "
          f"force {np.mean([k.force for k in kts])/st*1e3:.3f} | owned {[sl.n_own for sl in slabs]}", flush=True)
    for sl in slabs:
        sl.b.close()
    del slabs
    torch.cuda.empty_cache()
