#!/usr/bin/env python3
"""Workload study (CPU, numpy): what would idle lanes buy the force sweep if they evaluated pair
bodies for the lanes still busy (the owner lane adds the terms in canonical order)?

  python scripts/studies/helper_lanes.py DIR/sorted_4194304_100.npz [waves]

trips today = max_lane hits.  With helpers: per trip a busy lane retires 1 + min(cap, floor(idle/busy))
hits.  Printed for cap = 1, 3, 7 and for the ideal (mean hits)."""
import sys
import numpy as np

D = 100
H = np.float32(0.1)
H2 = H * H


def trips_with_helpers(h, cap):
    h = np.sort(h)[::-1].astype(np.int64).copy()
    t = 0
    while h[0] > 0:
        busy = int((h > 0).sum())
        per = 1 + min(cap, (64 - busy) // busy)
        # run until the next lane finishes
        nxt = h[busy - 1]                      # smallest remaining among busy lanes
        steps = -(-nxt // per)
        h[:busy] -= steps * per
        np.maximum(h, 0, out=h)
        t += steps
    return t


def main():
    path = sys.argv[1]
    nw = int(sys.argv[2]) if len(sys.argv) > 2 else 1500
    z = np.load(path)
    pos = z["pos"].astype(np.float32)
    c = np.clip((pos / H).astype(np.float32).astype(np.int64), 0, D - 1)
    key = c[:, 0] + D * c[:, 1] + D * D * c[:, 2]
    order = np.argsort(key, kind="stable")
    pos, key, c = pos[order], key[order], c[order]
    n = len(pos)
    cs = np.searchsorted(key, np.arange(D ** 3), side="left")
    ce = np.searchsorted(key, np.arange(D ** 3), side="right")
    rng = np.random.default_rng(3)
    waves = np.sort(rng.choice(n // 64, size=min(nw, n // 64), replace=False))
    tot = {"now": 0, 1: 0, 3: 0, 7: 0, "ideal": 0.0}
    for w in waves:
        i0 = w * 64
        P, C = pos[i0:i0 + 64], c[i0:i0 + 64]
        hits = np.zeros(64, np.int64)
        for r in range(9):
            dz, dy = r // 3 - 1, r % 3 - 1
            y, zc = C[:, 1] + dy, C[:, 2] + dz
            ok = (y >= 0) & (y < D) & (zc >= 0) & (zc < D)
            base = np.clip(y, 0, D - 1) * D + np.clip(zc, 0, D - 1) * D * D
            js = np.where(ok, cs[base + np.maximum(C[:, 0] - 1, 0)], 0)
            je = np.where(ok, ce[base + np.minimum(C[:, 0] + 1, D - 1)], 0)
            for l in range(64):
                if je[l] > js[l]:
                    d = P[l] - pos[js[l]:je[l]]
                    d2 = (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]
                    hits[l] += (d2 <= H2).sum()
        tot["now"] += hits.max()
        tot["ideal"] += hits.mean()
        for cap in (1, 3, 7):
            tot[cap] += trips_with_helpers(hits, cap)
    m = len(waves)
    print(f"{path}: trips/wave today {tot['now']/m:.1f} | one helper {tot[1]/m:.1f} ({tot[1]/tot['now']:.3f}) | "
          f"up to 3 helpers {tot[3]/m:.1f} ({tot[3]/tot['now']:.3f}) | up to 7 {tot[7]/m:.1f} ({tot[7]/tot['now']:.3f}) | "
          f"ideal {tot['ideal']/m:.1f} ({tot['ideal']/tot['now']:.3f})")


if __name__ == "__main__":
    main()
