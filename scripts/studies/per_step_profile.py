#!/usr/bin/env python3
"""Per-step kernel times of the headline run (and, with --count, pair tests / recorded hits /
pair bodies evaluated after the zero-pair filter): where the 100 steps' time sits.
usage: python scripts/studies/per_step_profile.py [--steps 100] [-n 4194304] [--count] [--every 5]
Prints one line per `--every` steps (averages over that window) and a JSON summary on the last line."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import cudafluidsimulator_amd as sph  # noqa: E402
from cudafluidsimulator_amd import _lib  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=100)
ap.add_argument("-n", type=int, default=4194304)
ap.add_argument("--every", type=int, default=5)
ap.add_argument("--count", action="store_true")
ap.add_argument("--math", default="strict")
ap.add_argument("--init", default="random")
ap.add_argument("--sweep", default="list")
args = ap.parse_args()

s = sph.default_settings(args.n, args.init == "random")
flags = _lib.SPH_FLAG_COUNT_PAIRS if args.count else 0
sim = sph.Simulator(s, flags=flags, math=args.math, sweep=args.sweep)
sim.setup()
t = sph.Times()
for _ in range(12):  # runtime settle (bench.py)
    sim.simulateAndTime(t)
sim.sync()
sim.setup()
sim.kernel_times(reset=True)
rows = []
acc = None
for k in range(1, args.steps + 1):
    sim.simulateAndTime(t)
    if k % args.every == 0 or k == args.steps:
        kt = sim.kernel_times(reset=True)
        st = max(int(kt.steps), 1)
        row = dict(step=k, grid=(kt.hash + kt.sort + kt.gather) / st * 1e3, density=kt.density / st * 1e3,
                   force=kt.force / st * 1e3)
        if args.count:
            row.update(tests=kt.pair_tests / st, bodies=kt.pair_hits / st)
        rows.append(row)
        print(" ".join(f"{a}={b:.4g}" if isinstance(b, float) else f"{a}={b}" for a, b in row.items()), flush=True)
sim.close()
print(json.dumps(rows))
