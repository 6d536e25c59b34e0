#!/usr/bin/env python3
"""Kernel times of steps 81..100 of the headline run (dense floor cells, pressure on) in
strict and fast math: is the force sweep VALU-bound there?  usage: python scripts/late_steps_ab.py"""
import os
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cudafluidsimulator_amd as sph

n = 4194304
s = sph.default_settings(n, True)
sim = sph.Simulator(s, flags=4)
sim.setup()
for _ in range(80):
    sim.simulate()
snap = os.path.join(tempfile.gettempdir(), "sph_step80.bin")
sim.save_state(snap)
sim.close()
for math in ("strict", "fast"):
    a = sph.Simulator(s, flags=4, math=math)
    a.load_state(snap)
    a.simulate()
    a.load_state(snap)
    a.kernel_times(reset=True)
    for _ in range(20):
        a.simulate()
    kt = a.kernel_times()
    print("%s: steps 81..100: density %.3f force %.3f grid %.3f ms/step" % (
        math, kt.density / 20 * 1e3, kt.force / 20 * 1e3, (kt.hash + kt.sort + kt.gather) / 20 * 1e3))
    a.close()
os.remove(snap)
