#!/usr/bin/env python3
"""Grid-build phase times (HIP events) with and without the position read-back beside them.
usage: [SPH_LIB_PATH=...] python scripts/studies/grid_build_split.py [n] [steps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import cudafluidsimulator_amd as sph
from cudafluidsimulator_amd import _lib

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4194304
K = int(sys.argv[2]) if len(sys.argv) > 2 else 20
s = sph.default_settings(n, True)
for label, flags in (("read-back on", 0), ("read-back off", _lib.SPH_FLAG_NO_READBACK)):
    for rep in range(2):
        sim = sph.Simulator(s, flags=flags)
        sim.setup()
        t = sph.Times()
        for _ in range(5):
            sim.simulateAndTime(t)
        sim.sync()
        sim.setup()
        sim.kernel_times(reset=True)
        for _ in range(K):
            sim.simulateAndTime(t)
        sim.sync()
        kt = sim.kernel_times()
        print("%-14s e0-e1 %.3f  sort %.3f  gather %.3f  density %.3f  force %.3f ms/step" % (
            label, kt.hash / K * 1e3, kt.sort / K * 1e3, kt.gather / K * 1e3, kt.density / K * 1e3, kt.force / K * 1e3))
        sim.close()
