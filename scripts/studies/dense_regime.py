#!/usr/bin/env python3
"""Workload study (CPU, numpy): where do the force sweep's trips go at late steps, and
would a per-wave choice between the shipped walk (each lane pops its own hit stream,
divergent gathers; trips = max_lane hits) and a run-synchronous walk over an LDS-staged
union (trips = sum_r max_lane hits_r, plus the staging) pay?

  python scripts/studies/dense_regime.py DIR/sorted_4194304_100.npz [waves]

Prints, by class of wave (number of cells its 64 particles span), the share of waves, of
hits and of whole-stream trips, and the run-synchronous trips of the same waves."""
import sys
import numpy as np

D = 100
H = np.float32(0.1)
H2 = H * H


def main():
    path = sys.argv[1]
    nw = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
    z = np.load(path)
    pos = z["pos"].astype(np.float32)
    c = np.clip((pos / H).astype(np.float32).astype(np.int64), 0, D - 1)
    key = c[:, 0] + D * c[:, 1] + D * D * c[:, 2]
    order = np.argsort(key, kind="stable")
    pos, key, c = pos[order], key[order], c[order]
    n = len(pos)
    cs = np.searchsorted(key, np.arange(D ** 3), side="left")
    ce = np.searchsorted(key, np.arange(D ** 3), side="right")
    rng = np.random.default_rng(2)
    waves = np.sort(rng.choice(n // 64, size=min(nw, n // 64), replace=False))
    rows = []
    for w in waves:
        i0 = w * 64
        P, C = pos[i0:i0 + 64], c[i0:i0 + 64]
        hits = np.zeros((64, 9), np.int64)
        ulen = np.zeros(9, np.int64)
        for r in range(9):
            dz, dy = r // 3 - 1, r % 3 - 1
            y, zc = C[:, 1] + dy, C[:, 2] + dz
            ok = (y >= 0) & (y < D) & (zc >= 0) & (zc < D)
            base = np.clip(y, 0, D - 1) * D + np.clip(zc, 0, D - 1) * D * D
            js = np.where(ok, cs[base + np.maximum(C[:, 0] - 1, 0)], 0)
            je = np.where(ok, ce[base + np.minimum(C[:, 0] + 1, D - 1)], 0)
            if ok.any():
                ulen[r] = je[ok].max() - js[ok].min()
            for l in range(64):
                if je[l] > js[l]:
                    d = P[l] - pos[js[l]:je[l]]
                    d2 = (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]
                    hits[l, r] = (d2 <= H2).sum() - (1 if r == 4 else 0)   # not itself
        ncell = len(np.unique(key[i0:i0 + 64]))
        rows.append((ncell, hits.sum(), hits.sum(axis=1).max(), hits.max(axis=0).sum(), ulen.sum(), ulen.max()))
    a = np.array(rows, dtype=np.float64)
    tot_h, tot_t = a[:, 1].sum(), a[:, 2].sum()
    print(f"{path}: {len(a)} waves, hits/particle {tot_h/len(a)/64:.1f}, whole-stream trips/wave {tot_t/len(a):.1f}, "
          f"run-synchronous {a[:,3].sum()/len(a):.1f}")
    print("cells spanned | waves % | hits % | trips % | trips/wave | run-sync trips/wave | ratio | union records/wave | longest run")
    for lo, hi in ((1, 1), (2, 2), (3, 4), (5, 8), (9, 16), (17, 64)):
        m = (a[:, 0] >= lo) & (a[:, 0] <= hi)
        if not m.any():
            continue
        s = a[m]
        print(f"{lo:3d}-{hi:<3d}      | {100*m.mean():6.1f} | {100*s[:,1].sum()/tot_h:6.1f} | {100*s[:,2].sum()/tot_t:6.1f} | "
              f"{s[:,2].mean():8.1f} | {s[:,3].mean():8.1f} | {s[:,3].sum()/s[:,2].sum():.2f} | {s[:,4].mean():8.0f} | {s[:,5].mean():6.0f}")
    # a per-wave choice: run-synchronous LDS walk costs `c` of a gather trip per trip plus the staging
    for cl in (0.35, 0.5, 0.7):
        stage = a[:, 4] / 64 * 0.25     # coalesced 32-B loads, 64 records per wave instruction pair
        alt = a[:, 3] * cl + stage
        best = np.minimum(a[:, 2], alt)
        print(f"LDS trip = {cl:.2f} gather trips: hybrid {best.sum()/tot_t:.3f} of today's trips "
              f"({100*(alt < a[:,2]).mean():.1f} % of waves switch)")


if __name__ == "__main__":
    main()
