#!/usr/bin/env python3
"""Why is the density sweep slower in steps 1..20 than in steps 25..45 of the same free fall (same
pair tests)?  Hypothesis: the GPU idles ~100 ms while the host prepares and uploads the initial
condition, and its clocks take tens of milliseconds of work to come back.  Test: per-step kernel
times of steps 1..40 (a) right after setup() (the bench's situation), (b) of a second simulator
that was set up EARLIER and starts right behind 40 busy steps of the first one.
usage: python scripts/studies/clock_ramp.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import cudafluidsimulator_amd as sph  # noqa: E402

n = 4194304
s = sph.default_settings(n, True)
a = sph.Simulator(s)
b = sph.Simulator(s)
b.setup()
a.setup()
t = sph.Times()


def run(sim, label, steps=40, every=5):
    sim.kernel_times(reset=True)
    for k in range(1, steps + 1):
        sim.simulateAndTime(t)
        if k % every == 0:
            kt = sim.kernel_times(reset=True)
            st = max(int(kt.steps), 1)
            print(f"{label} step {k}: grid {1e3*(kt.hash+kt.sort+kt.gather)/st:.3f} density {1e3*kt.density/st:.3f} "
                  f"force {1e3*kt.force/st:.3f}", flush=True)


run(a, "A (right after setup)")
run(b, "B (set up earlier, starts behind A's 40 busy steps)")
a.setup()
run(a, "A again (right after another setup)")
a.close()
b.close()
