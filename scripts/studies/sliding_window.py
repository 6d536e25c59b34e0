#!/usr/bin/env python3
"""Force sweep: would a window that SLIDES along the sorted stream with the wave's lanes serve more hits from
LDS than the shipped fixed window around the wave's own particles?  CPU study on an oracle state (npz with
pos, vel, rho): per sampled wave of 64 consecutive sorted particles, every lane's hit list in canonical order
(ascending sorted index over the nine runs), trips in lock-step (trip t = every live lane's t-th hit).
  shipped : records [w0, w0 + 160) around the wave's own particles
  sliding : records [ws, ws + W); whenever fewer than `thr` of the live lanes find their hit inside, the
            window is re-staged at the q-quantile of the live lanes' current indices
usage: python scripts/studies/sliding_window.py state.npz [waves] [min cell occupancy] [subcells per axis]
  subcells per axis S > 1: particles inside a cell ordered by an S x S x S sub-cell index (x fastest) -- what a
  sub-cell key order would give: lanes of a wave spatially coherent.  Also printed: lane efficiency (mean / max
  hits per wave) and distinct 128-byte lines per trip among the lanes outside the shipped window."""
import sys
import numpy as np

D, H = 100, np.float32(0.1)
z = np.load(sys.argv[1])
nw = int(sys.argv[2]) if len(sys.argv) > 2 else 150
minocc = int(sys.argv[3]) if len(sys.argv) > 3 else 0
SUB = int(sys.argv[4]) if len(sys.argv) > 4 else 1
pos, vel, rho = z["pos"], z["vel"], z["rho"]
n = len(pos)
c = np.clip((pos / H).astype(np.int64), 0, D - 1)
key = c[:, 0] + D * c[:, 1] + D * D * c[:, 2]
if SUB > 1:
    f = np.clip(((pos / H - c) * SUB).astype(np.int64), 0, SUB - 1)
    key = key * (SUB ** 3) + f[:, 0] + SUB * f[:, 1] + SUB * SUB * f[:, 2]
order = np.argsort(key, kind="stable")
if SUB > 1:
    key = key // (SUB ** 3)
pos, vel, rho, key, c = pos[order], vel[order], rho[order], key[order], c[order]
start = np.searchsorted(key, np.arange(D ** 3), side="left")
end = np.searchsorted(key, np.arange(D ** 3), side="right")
occ = (end - start)[key]
v = vel.view([("a", "f4"), ("b", "f4"), ("c", "f4")]).ravel()
u, cnt = np.unique(v, return_counts=True)
quiet = (v == u[cnt.argmax()]) & (rho <= 1000.0)
rng = np.random.default_rng(5)
cands = np.arange(n // 64)
if minocc:
    cands = cands[occ[cands * 64] >= minocc]
waves = np.sort(rng.choice(cands, size=min(nw, len(cands)), replace=False))


def hit_lists(w):
    out = []
    for i in range(w * 64, w * 64 + 64):
        ci, hs = c[i], []
        for dz in (-1, 0, 1):
            for dy in (-1, 0, 1):
                y, zz = ci[1] + dy, ci[2] + dz
                if not (0 <= y < D and 0 <= zz < D):
                    continue
                a = start[max(ci[0] - 1, 0) + D * y + D * D * zz]
                b = end[min(ci[0] + 1, D - 1) + D * y + D * D * zz]
                if b > a:
                    d = pos[i] - pos[a:b]
                    hit = ((d * d).sum(axis=1) <= H * H) & ~(quiet[a:b] & quiet[i])
                    hs.append(a + np.nonzero(hit)[0])
        out.append(np.concatenate(hs) if hs else np.zeros(0, np.int64))
    return out


res = {}
configs = [("shipped 160", None), ("sliding W=160 thr=1/2 q=0.25", (160, 0.5, 0.25)), ("sliding W=160 thr=1/4 q=0.25", (160, 0.25, 0.25)),
           ("sliding W=160 thr=1/4 q=0.1", (160, 0.25, 0.1)), ("sliding W=256 thr=1/4 q=0.25", (256, 0.25, 0.25)),
           ("sliding W=96 thr=1/4 q=0.25", (96, 0.25, 0.25))]
tot = {k: [0, 0, 0, 0] for k, _ in configs}   # hits, served, restages, trips
eff = [0, 0]      # sum of mean hits, sum of max hits
lines = [0, 0]    # distinct lines among out-of-window lanes, trips
for w in waves:
    L = hit_lists(w)
    T = max(len(x) for x in L)
    if T == 0:
        continue
    J = np.full((64, T), -1, np.int64)
    for l, x in enumerate(L):
        J[l, :len(x)] = x
    live = J >= 0
    eff[0] += live.sum() / 64.0
    eff[1] += T
    w0_ = max(w * 64 - 48, 0)
    for t in range(0, T, 7):
        jl = J[live[:, t], t]
        jl = jl[(jl < w0_) | (jl >= w0_ + 160)]
        lines[0] += len(np.unique(jl // 4))
        lines[1] += 1
    for name, cfg in configs:
        if cfg is None:
            w0 = max(w * 64 - 48, 0)
            served = (live & (J >= w0) & (J < w0 + 160)).sum()
            rest = 0
        else:
            W, thr, q = cfg
            ws, served, rest = -10 ** 9, 0, 0
            for t in range(T):
                jl = J[live[:, t], t]
                inw = (jl >= ws) & (jl < ws + W)
                if inw.sum() < thr * len(jl):
                    ws = int(np.quantile(jl, q)) - 8
                    rest += 1
                    inw = (jl >= ws) & (jl < ws + W)
                served += int(inw.sum())
        a = tot[name]
        a[0] += int(live.sum()); a[1] += int(served); a[2] += rest; a[3] += T
print(f"{sys.argv[1]}: {len(waves)} waves" + (f" starting in cells of >= {minocc}" if minocc else "") +
      (f", particles ordered by {SUB}^3 sub-cells inside a cell" if SUB > 1 else "") +
      f"; lane efficiency {eff[0]/max(eff[1],1):.3f}; distinct lines per trip outside the window {lines[0]/max(lines[1],1):.1f}")
for name, _ in configs:
    h, s, r, t = tot[name]
    print(f"  {name:32s}: served from LDS {100*s/max(h,1):5.1f} % of {h/len(waves):7.0f} pair bodies per wave; "
          f"{r/len(waves):5.1f} re-stagings per wave ({t/len(waves):6.0f} trips)")
