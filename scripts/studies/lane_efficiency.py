#!/usr/bin/env python3
"""Workload study (CPU, numpy): how many lock-step trips a 64-lane wave needs for the
density and force sweeps under different walk strategies, on key-sorted particle
states dumped by tests/golden/make_golden.py --full --dump-dir DIR.

  python scripts/studies/lane_efficiency.py DIR/sorted_4194304_60.npz [waves]

Strategies (one lane per particle, 64 consecutive sorted particles per wave):
  density  per-run   : sum_r max_lane ceil(len_r/4)        (round 1: lock-step per run)
           concat    : max_lane sum_r ceil(len_r/4)        (lanes run through their 9 runs back to back)
           concat+trim: the same after dropping x-1 / x+1 cells (and whole runs) that the
                       particle's distance to the cell faces proves out of reach
           ideal     : sum_lane sum_r len_r / 64 / 4
  force    whole     : max_lane hits                       (round 1: per-lane hit stream)
           per-run   : sum_r max_lane hits_r               (LDS staging run by run)
           per-layer : sum_dz max_lane hits_dz             (LDS staging three runs at a time)
           ideal     : mean hits
"""
import sys
import numpy as np

D = 100
H = np.float32(0.1)
H2 = H * H


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
    rng = np.random.default_rng(1)
    waves = np.sort(rng.choice(n // 64, size=min(nw, n // 64), replace=False))
    acc = {k: 0.0 for k in ("d_run", "d_cat", "d_trim", "d_ideal", "f_whole", "f_run", "f_layer", "f_ideal",
                             "cand", "cand_trim", "hits")}
    for w in waves:
        i0 = w * 64
        P = pos[i0:i0 + 64]
        C = c[i0:i0 + 64]
        len4 = np.zeros((64, 9), np.int64)
        len4t = np.zeros((64, 9), np.int64)
        hits = np.zeros((64, 9), np.int64)
        cand = 0
        candt = 0
        for r in range(9):
            dz, dy = r // 3 - 1, r % 3 - 1
            y, zc = C[:, 1] + dy, C[:, 2] + dz
            ok = (y >= 0) & (y < D) & (zc >= 0) & (zc < D)
            base = np.clip(y, 0, D - 1) * D + np.clip(zc, 0, D - 1) * D * D
            x0 = np.maximum(C[:, 0] - 1, 0)
            x1 = np.minimum(C[:, 0] + 1, D - 1)
            js = np.where(ok, cs[base + x0], 0)
            je = np.where(ok, ce[base + x1], 0)
            # geometric trimming: squared distance from the particle to the neighbour cell's box
            fy = np.where(dy < 0, P[:, 1] - C[:, 1] * H, np.where(dy > 0, (C[:, 1] + 1) * H - P[:, 1], 0))
            fz = np.where(dz < 0, P[:, 2] - C[:, 2] * H, np.where(dz > 0, (C[:, 2] + 1) * H - P[:, 2], 0))
            fxm = P[:, 0] - C[:, 0] * H          # to the x-1 cell
            fxp = (C[:, 0] + 1) * H - P[:, 0]    # to the x+1 cell
            g = fy * fy + fz * fz
            lim = H2 * np.float32(1.001)
            jst = np.where(g + fxm * fxm > lim, np.where(ok, cs[base + C[:, 0]], 0), js)
            jet = np.where(g + fxp * fxp > lim, np.where(ok, ce[base + C[:, 0]], 0), je)
            drop = g > lim
            jst = np.where(drop, 0, jst)
            jet = np.where(drop, 0, jet)
            ln = je - js
            lnt = np.maximum(jet - jst, 0)
            len4[:, r] = (ln + 3) // 4
            len4t[:, r] = (lnt + 3) // 4
            cand += ln.sum()
            candt += lnt.sum()
            for l in range(64):
                if ln[l] > 0:
                    d = P[l] - pos[js[l]:je[l]]
                    d2 = (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]
                    hm = d2 <= H2
                    hits[l, r] = hm.sum()
                    if lnt[l] < ln[l]:  # trimming must never drop a hit
                        a, b = jst[l] - js[l], jet[l] - js[l]
                        keep = np.zeros(len(hm), bool)
                        if b > a:
                            keep[a:b] = True
                        assert not (hm & ~keep).any(), "trim dropped a hit"
        acc["d_run"] += len4.max(axis=0).sum()
        acc["d_cat"] += len4.sum(axis=1).max()
        acc["d_trim"] += len4t.sum(axis=1).max()
        acc["d_ideal"] += cand / 64 / 4
        acc["cand"] += cand
        acc["cand_trim"] += candt
        acc["hits"] += hits.sum()
        acc["f_whole"] += hits.sum(axis=1).max()
        acc["f_run"] += hits.max(axis=0).sum()
        acc["f_layer"] += sum(hits[:, 3 * g:3 * g + 3].sum(axis=1).max() for g in range(3))
        acc["f_ideal"] += hits.sum() / 64
    m = len(waves)
    print(f"{path}: {m} waves sampled, candidates/particle {acc['cand']/m/64:.1f} "
          f"(trimmed {acc['cand_trim']/m/64:.1f}), hits/particle {acc['hits']/m/64:.1f} "
          f"({100*acc['hits']/acc['cand']:.1f} % of candidates)")
    print("density trips/wave: per-run %.1f | concat %.1f | concat+trim %.1f | ideal %.1f" %
          (acc["d_run"] / m, acc["d_cat"] / m, acc["d_trim"] / m, acc["d_ideal"] / m))
    print("force   trips/wave: whole-stream %.1f | per-run %.1f | per-layer(3 runs) %.1f | ideal %.1f" %
          (acc["f_whole"] / m, acc["f_run"] / m, acc["f_layer"] / m, acc["f_ideal"] / m))


if __name__ == "__main__":
    main()
