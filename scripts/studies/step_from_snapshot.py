#!/usr/bin/env python3
"""Kernel times of ONE step from a snapshot of the headline run, per library variant -- for perf-only builds
whose results are wrong by construction and must not feed the next step.
  python scripts/studies/step_from_snapshot.py make STEP path     (default library: STEP steps, then save_state)
  python scripts/studies/step_from_snapshot.py time path [reps]   (SPH_LIB_PATH selects the variant)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import cudafluidsimulator_amd as sph

n = 4194304
if sys.argv[1] == "make":
    step, path = int(sys.argv[2]), sys.argv[3]
    sim = sph.Simulator(sph.default_settings(n, True))
    sim.setup()
    for _ in range(step):
        sim.simulate()
    sim.save_state(path)
    sim.close()
    print("saved step", step, "to", path)
else:
    path = sys.argv[2]
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    out = []
    for _ in range(reps):
        sim = sph.Simulator(sph.default_settings(n, True))
        sim.load_state(path)
        sim.kernel_times(reset=True)
        sim.simulate()
        kt = sim.kernel_times()
        out.append((kt.density * 1e3, kt.force * 1e3))
        sim.close()
    print(os.path.basename(os.environ.get("SPH_LIB_PATH", "main")), "one step from", os.path.basename(path),
          " ".join("density %.3f force %.3f |" % o for o in out), flush=True)
