#!/usr/bin/env python3
"""Which particles carry the force sweep's pair bodies once the zero-pair filter has dropped the
quiet-quiet pairs?  (Would a 3-D blocked cell order -- whose pay-off is a smaller neighbourhood for
waves that span MANY sparse cells -- still have anything to shrink?)  CPU study on oracle states
(npz with pos, vel, rho), sampled particles.
  python scripts/studies/bodies_by_occupancy.py state.npz [samples]"""
import sys
import numpy as np

D, H = 100, np.float32(0.1)
z = np.load(sys.argv[1])
ns = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
pos, vel, rho = z["pos"], z["vel"], z["rho"]
n = len(pos)
c = np.clip((pos / H).astype(np.int64), 0, D - 1)
key = c[:, 0] + D * c[:, 1] + D * D * c[:, 2]
order = np.argsort(key, kind="stable")
pos, vel, rho, key, c = pos[order], vel[order], rho[order], key[order], c[order]
start = np.searchsorted(key, np.arange(D ** 3), side="left")
end = np.searchsorted(key, np.arange(D ** 3), side="right")
occ = (end - start)[key]
v = vel.view([("a", "f4"), ("b", "f4"), ("c", "f4")]).ravel()
u, cnt = np.unique(v, return_counts=True)
vref = u[cnt.argmax()]
quiet = (v == vref) & (rho <= 1000.0)
rng = np.random.default_rng(1)
S = rng.choice(n, ns, replace=False)
bodies = np.zeros(ns)
hits = np.zeros(ns)
for q, i in enumerate(S):
    ci = c[i]
    tot = kept = 0
    for dz in (-1, 0, 1):
        for dy in (-1, 0, 1):
            y, zz = ci[1] + dy, ci[2] + dz
            if not (0 <= y < D and 0 <= zz < D):
                continue
            x0, x1 = max(ci[0] - 1, 0), min(ci[0] + 1, D - 1)
            a, b = start[x0 + D * y + D * D * zz], end[x1 + D * y + D * D * zz]
            if b > a:
                d = pos[i] - pos[a:b]
                hit = (d * d).sum(axis=1) <= H * H
                tot += int(hit.sum())
                kept += int((hit & ~(quiet[a:b] & quiet[i])).sum())
    hits[q], bodies[q] = tot, kept
o = occ[S]
print(f"{sys.argv[1]}: quiet particles {100*quiet.mean():.1f} %, recorded hits/particle {hits.mean():.1f}, "
      f"pair bodies/particle after the filter {bodies.mean():.1f} ({100*bodies.sum()/hits.sum():.0f} % of the hits)")
for lo, hi in ((1, 7), (8, 15), (16, 31), (32, 63), (64, 10 ** 9)):
    m = (o >= lo) & (o <= hi)
    print(f"  particles in cells of {lo:3d}..{'' if hi > 10**6 else hi:<4} : {100*m.mean():5.1f} % of the particles, "
          f"{100*bodies[m].sum()/max(bodies.sum(),1):5.1f} % of the pair bodies, {64/np.maximum(o[m],1).mean() if m.any() else 0:5.1f} cells per 64-lane wave")
