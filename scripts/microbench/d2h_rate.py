#!/usr/bin/env python3
"""Back-to-back device-to-host copies of the read-back's size on one stream: what one copy really costs
(the floor of a step that reads its positions back, simulator.cu:479-480).  usage: python scripts/microbench/d2h_rate.py"""
import time
import torch

n = 4194304 * 3
dev = torch.empty(n, dtype=torch.float32, device="cuda")
host = torch.empty(n, dtype=torch.float32).pin_memory()
s = torch.cuda.Stream()
for reps in (1, 50):
    with torch.cuda.stream(s):
        for _ in range(5):
            host.copy_(dev, non_blocking=True)
        s.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            host.copy_(dev, non_blocking=True)
        s.synchronize()
        dt = time.perf_counter() - t0
    print(f"{reps} copies of {n*4/1e6:.1f} MB back to back: {dt/reps*1e3:.3f} ms each, {n*4/1e9/(dt/reps):.1f} GB/s")
