#!/usr/bin/env python3
"""One-off ~7 ms stalls in the first steps of a process: per-step wall times of simulateAndTime
in bench.py's sequence (setup, W warm-up steps, setup, K steps), pieces switched by environment:
TORCH=1 import torch first | TSYNC=1 torch.cuda.synchronize() before the loop | RESET=1
kernel_times(reset) before the loop | NORB=1 no read-back | W=<warm-up steps>
usage: python scripts/studies/early_stall.py [n] [steps]"""
import os
if os.environ.get("TORCH"):
    import torch
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import cudafluidsimulator_amd as sph
from cudafluidsimulator_amd import _lib

n = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
K = int(sys.argv[2]) if len(sys.argv) > 2 else 10
W = int(os.environ.get("W", "2"))
flags = _lib.SPH_FLAG_NO_READBACK if os.environ.get("NORB") else 0
if os.environ.get("TSYNC"):
    torch.cuda.init()
    torch.cuda.synchronize()
sim = sph.Simulator(sph.default_settings(n, True), flags=flags, device=0)
sim.setup()
t = sph.Times()
w = []
for i in range(W):
    t0 = time.perf_counter()
    sim.simulateAndTime(t)
    w.append((time.perf_counter() - t0) * 1e3)
sim.sync()
sim.setup()
if os.environ.get("RESET"):
    sim.kernel_times(reset=True)
t = sph.Times()
if os.environ.get("TSYNC"):
    torch.cuda.synchronize()
w.append(-1)
for i in range(K):
    t0 = time.perf_counter()
    sim.simulateAndTime(t)
    w.append((time.perf_counter() - t0) * 1e3)
print(" ".join("%.2f" % x if x >= 0 else "|" for x in w))
sim.close()
