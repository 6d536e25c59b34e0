#!/usr/bin/env python3
"""Hit stream: bytes per row of the shipped encoding ((first candidate, 32-bit mask) pairs, EMPTY words skipped,
two pairs per 16-byte quad, zero terminator) against a "run descriptor + mask words" encoding (9 descriptors per
row, then every 32-candidate word of every run, no bases).  CPU study on oracle states.
  python scripts/studies/stream_encoding.py state.npz [...]"""
import sys, numpy as np
D, H = 100, np.float32(0.1)
for path in sys.argv[1:]:
    z = np.load(path); pos = z["pos"]; n = len(pos)
    c = np.clip((pos / H).astype(np.int64), 0, D - 1)
    key = c[:, 0] + D * c[:, 1] + D * D * c[:, 2]
    order = np.argsort(key, kind="stable"); pos, key, c = pos[order], key[order], c[order]
    start = np.searchsorted(key, np.arange(D ** 3), side="left"); end = np.searchsorted(key, np.arange(D ** 3), side="right")
    rng = np.random.default_rng(3); rows = rng.choice(n, 4000, replace=False)
    now = new = cand = 0
    for i in rows:
        ci = c[i]; words_new = 0; nonempty = 0
        for dz in (-1, 0, 1):
            for dy in (-1, 0, 1):
                y, zz = ci[1] + dy, ci[2] + dz
                if not (0 <= y < D and 0 <= zz < D): continue
                a = start[max(ci[0] - 1, 0) + D * y + D * D * zz]; b = end[min(ci[0] + 1, D - 1) + D * y + D * D * zz]
                if b <= a: continue
                d = pos[i] - pos[a:b]; hit = (d * d).sum(axis=1) <= H * H
                nw = (b - a + 31) // 32; words_new += nw; cand += b - a
                pad = np.zeros(nw * 32, bool); pad[:b - a] = hit
                nonempty += int(pad.reshape(nw, 32).any(axis=1).sum())
        now += 16 * ((nonempty + 1 + 1) // 2)          # pairs of 8 B, two per quad, + terminator
        new += 36 + 16 * ((words_new + 3) // 4)
    print(path, "candidates/row %.0f  stream bytes/row: now %.0f, run descriptors + mask words %.0f (%.0f %%)" % (cand / len(rows), now / len(rows), new / len(rows), 100 * new / now))
