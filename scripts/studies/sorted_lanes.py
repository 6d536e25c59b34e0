#!/usr/bin/env python3
"""Force sweep: trips per wave = the largest pair-body count among its 64 lanes.  If a workgroup took G
consecutive 64-row waves and dealt its rows to lanes SORTED by their hit counts (each lane still walks its own
row's stream in canonical order: same results), how many trips would be left?  CPU study on an oracle state.
  python scripts/studies/sorted_lanes.py state.npz [groups] [G]"""
import sys
import numpy as np

D, H = 100, np.float32(0.1)
z = np.load(sys.argv[1])
ng = int(sys.argv[2]) if len(sys.argv) > 2 else 120
G = int(sys.argv[3]) if len(sys.argv) > 3 else 4
pos, vel, rho = z["pos"], z["vel"], z["rho"]
n = len(pos)
c = np.clip((pos / H).astype(np.int64), 0, D - 1)
key = c[:, 0] + D * c[:, 1] + D * D * c[:, 2]
order = np.argsort(key, kind="stable")
pos, vel, rho, key, c = pos[order], vel[order], rho[order], key[order], c[order]
start = np.searchsorted(key, np.arange(D ** 3), side="left")
end = np.searchsorted(key, np.arange(D ** 3), side="right")
v = vel.view([("a", "f4"), ("b", "f4"), ("c", "f4")]).ravel()
u, cnt = np.unique(v, return_counts=True)
quiet = (v == u[cnt.argmax()]) & (rho <= 1000.0)
rng = np.random.default_rng(9)
groups = np.sort(rng.choice(n // (64 * G), size=ng, replace=False))
tot = dict(bodies=0, trips_now=0, trips_sorted_hits=0, trips_sorted_bodies=0, g2=0, g4=0, g8=0)
for g in groups:
    i0 = g * 64 * G
    hits = np.zeros(64 * G, np.int64)
    bodies = np.zeros(64 * G, np.int64)
    for q, i in enumerate(range(i0, i0 + 64 * G)):
        ci = c[i]
        for dz in (-1, 0, 1):
            for dy in (-1, 0, 1):
                y, zz = ci[1] + dy, ci[2] + dz
                if not (0 <= y < D and 0 <= zz < D):
                    continue
                a = start[max(ci[0] - 1, 0) + D * y + D * D * zz]
                b = end[min(ci[0] + 1, D - 1) + D * y + D * D * zz]
                if b > a:
                    d = pos[i] - pos[a:b]
                    hit = (d * d).sum(axis=1) <= H * H
                    hits[q] += int(hit.sum())
                    bodies[q] += int((hit & ~(quiet[a:b] & quiet[i])).sum())
    tot["bodies"] += int(bodies.sum())
    tot["trips_now"] += int(bodies.reshape(G, 64).max(axis=1).sum())
    tot["trips_sorted_hits"] += int(bodies[np.argsort(hits, kind="stable")].reshape(G, 64).max(axis=1).sum())
    tot["trips_sorted_bodies"] += int(np.sort(bodies).reshape(G, 64).max(axis=1).sum())
    for gs in (2, 4, 8):  # rows dealt in groups of gs ADJACENT rows (sorted by the group's largest hit count)
        o = np.argsort(hits.reshape(-1, gs).max(axis=1), kind="stable")
        tot["g%d" % gs] += int(bodies.reshape(-1, gs)[o].reshape(G, 64).max(axis=1).sum())
ideal = tot["bodies"] / 64
print(f"{sys.argv[1]}: {ng} groups of {G} waves; lane efficiency (bodies / (64 x trips)): today {ideal/tot['trips_now']:.3f}, rows dealt to lanes "
      f"sorted by recorded hits {ideal/tot['trips_sorted_hits']:.3f}, sorted by pair bodies after the filter {ideal/tot['trips_sorted_bodies']:.3f}; "
      f"groups of 2 / 4 / 8 adjacent rows {ideal/tot['g2']:.3f} / {ideal/tot['g4']:.3f} / {ideal/tot['g8']:.3f}; "
      f"trips {tot['trips_now']} -> {tot['trips_sorted_hits']} ({100*(1-tot['trips_sorted_hits']/tot['trips_now']):.1f} % fewer)")
