#!/usr/bin/env python3
"""Workload study (CPU, numpy) for the force sweep's MOVING LDS window: what fraction of
the hit records a wave pops would be served from LDS if the wave kept the unions of TWO
consecutive runs resident (capacity WCAP records each) and moved on when the majority of
its lanes has left the older run.  Lanes never wait: a hit outside the resident windows is
gathered from global memory as today.

  python scripts/studies/moving_window.py DIR/sorted_4194304_60.npz [WCAP] [waves]
"""
import sys
import numpy as np

D = 100
H = np.float32(0.1)
H2 = H * H


def main():
    z = np.load(sys.argv[1])
    wcap = int(sys.argv[2]) if len(sys.argv) > 2 else 96
    nw = int(sys.argv[3]) if len(sys.argv) > 3 else 300
    pos = z["pos"].astype(np.float32)
    c = np.clip((pos / H).astype(np.float32).astype(np.int64), 0, D - 1)
    key = c[:, 0] + D * c[:, 1] + D * D * c[:, 2]
    order = np.argsort(key, kind="stable")
    pos, key, c = pos[order], key[order], c[order]
    n = len(pos)
    cs = np.searchsorted(key, np.arange(D ** 3), side="left")
    ce = np.searchsorted(key, np.arange(D ** 3), side="right")
    rng = np.random.default_rng(1)
    waves = np.sort(rng.choice(n // 64, size=min(nw, n // 64), replace=False))
    tot = lds2 = lds1 = own = 0
    trips = 0
    for w in waves:
        i0 = w * 64
        P, C = pos[i0:i0 + 64], c[i0:i0 + 64]
        hits = [[] for _ in range(64)]          # per lane: list of (run, j)
        u0 = np.zeros(9, np.int64)
        u1 = np.zeros(9, np.int64)
        for r in range(9):
            dz, dy = r // 3 - 1, r % 3 - 1
            y, zc = C[:, 1] + dy, C[:, 2] + dz
            ok = (y >= 0) & (y < D) & (zc >= 0) & (zc < D)
            base = np.clip(y, 0, D - 1) * D + np.clip(zc, 0, D - 1) * D * D
            js = np.where(ok, cs[base + np.maximum(C[:, 0] - 1, 0)], 0)
            je = np.where(ok, ce[base + np.minimum(C[:, 0] + 1, D - 1)], 0)
            ne = je > js
            if ne.any():
                u0[r], u1[r] = js[ne].min(), je[ne].max()
            for l in range(64):
                if je[l] > js[l]:
                    d = P[l] - pos[js[l]:je[l]]
                    d2 = (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]
                    for j in np.nonzero(d2 <= H2)[0]:
                        hits[l].append((r, js[l] + j))
        nh = np.array([len(h) for h in hits])
        T = nh.max()
        trips += T
        a = 0                                    # window A = run a, window B = run a+1
        for t in range(T):
            live = [l for l in range(64) if t < nh[l]]
            beyond = 0
            for l in live:
                r, j = hits[l][t]
                tot += 1
                inA = a <= 8 and u0[a] <= j < min(u1[a], u0[a] + wcap)
                inB = a + 1 <= 8 and u0[a + 1] <= j < min(u1[a + 1], u0[a + 1] + wcap)
                lds2 += inA or inB
                lds1 += inA
                own += abs(j - (i0 + 32)) < 80      # today's fixed 160-record window
                if r > a:
                    beyond += 1
            while a < 8 and beyond * 2 > len(live):  # the majority has left run a
                a += 1
                beyond = sum(1 for l in live if hits[l][t][0] > a)
    print(f"{sys.argv[1]} WCAP={wcap}: hits served from LDS: two moving windows {100*lds2/tot:.1f} % "
          f"(older window alone {100*lds1/tot:.1f} %), today's fixed own-row window {100*own/tot:.1f} %; "
          f"trips/wave {trips/len(waves):.1f}")


if __name__ == "__main__":
    main()
