#!/usr/bin/env python3
"""Why the fast sweeps are built on the flattened key (x + y D + z D^2) and not on the Morton key
(BASELINE config 3's "z-index sort"): what a 64-lane wave would have to stage for its neighbourhood
under either ordering.  CPU study on a particle state (numpy).

  python scripts/studies/morton_fragmentation.py [state.npz with pos] [waves]

Per wave of 64 consecutive sorted particles: the set of cells its lanes' 27-cell neighbourhoods touch,
the number of CONTIGUOUS ranges of the sorted stream that set forms (= coalesced staging loads /
range descriptors), the records in it (= LDS footprint of staging the whole neighbourhood at 16 B per
position), and the lock-step trip count of a walk that visits the ranges one after the other (every
lane waits for the lane with the most candidates in the current range)."""
import os
import sys
import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
D = 100
H = np.float32(0.1)


def spread(v):
    out = np.zeros_like(v)
    for b in range(7):
        out |= ((v >> b) & 1) << (3 * b)
    return out


def morton(cx, cy, cz):
    return spread(cx) | (spread(cy) << 1) | (spread(cz) << 2)


def load(path):
    if path and os.path.exists(path):
        return np.load(path)["pos"].astype(np.float32)
    from oracle import oracle as O
    sim = O.OracleSim(4194304, True)
    sim.setup()
    return sim.download()["pos"]


def study(pos, keyfn, name, nw, rng):
    c = np.clip((pos / H).astype(np.float32).astype(np.int64), 0, D - 1)
    key = keyfn(c[:, 0], c[:, 1], c[:, 2])
    order = np.argsort(key, kind="stable")
    key, c = key[order], c[order]
    n = len(key)
    nk = (1 << 21) if keyfn is morton else D ** 3
    start = np.searchsorted(key, np.arange(nk), side="left")
    end = np.searchsorted(key, np.arange(nk), side="right")
    waves = np.sort(rng.choice(n // 64, size=min(nw, n // 64), replace=False))
    offs = np.array([(dx, dy, dz) for dz in (-1, 0, 1) for dy in (-1, 0, 1) for dx in (-1, 0, 1)])
    res = []
    for w in waves:
        C = c[w * 64:(w + 1) * 64]
        nb = C[:, None, :] + offs[None, :, :]                       # 64 x 27 x 3
        ok = ((nb >= 0) & (nb < D)).all(axis=2)
        k = keyfn(np.clip(nb[..., 0], 0, D - 1), np.clip(nb[..., 1], 0, D - 1), np.clip(nb[..., 2], 0, D - 1))
        cells = np.unique(k[ok])
        cells = cells[end[cells] > start[cells]]
        s, e = start[cells], end[cells]
        # contiguous ranges of the sorted stream (adjacent occupied cells merge; empty cells in between too)
        o = np.argsort(s)
        s, e = s[o], e[o]
        breaks = np.concatenate([[True], s[1:] != e[:-1]])
        nranges = int(breaks.sum())
        records = int((e - s).sum())
        # lock-step walk, one contiguous range at a time: trips = sum over ranges of the largest per-lane count
        rid = np.cumsum(breaks) - 1
        cell_range = dict(zip(cells[o].tolist(), rid.tolist()))
        cnt = np.zeros((64, nranges), np.int64)
        for l in range(64):
            for q in range(27):
                if ok[l, q]:
                    kk = int(k[l, q])
                    if kk in cell_range:
                        cnt[l, cell_range[kk]] += end[kk] - start[kk]
        res.append((nranges, records, int(cnt.max(axis=0).sum()), int(cnt.sum(axis=1).max()), float(cnt.sum(axis=1).mean())))
    a = np.array(res, dtype=np.float64)
    print(f"{name:10s}: contiguous ranges per wave {a[:,0].mean():6.1f} (max {a[:,0].max():.0f}) | records in the "
          f"neighbourhood {a[:,1].mean():7.0f} = {a[:,1].mean()*16/1024:5.1f} KB of positions | lock-step trips, range by "
          f"range {a[:,2].mean():7.0f} vs longest lane {a[:,3].mean():6.0f} (mean lane {a[:,4].mean():6.0f})")


def main():
    path = sys.argv[1] if len(sys.argv) > 1 else None
    nw = int(sys.argv[2]) if len(sys.argv) > 2 else 400
    pos = load(path)
    print(f"{path or 'reference -i random start, n = 4194304'}: {len(pos)} particles, {nw} sampled waves")
    study(pos, lambda x, y, z: x + D * y + D * D * z, "flattened", nw, np.random.default_rng(3))
    study(pos, morton, "morton", nw, np.random.default_rng(3))


if __name__ == "__main__":
    main()
