#!/usr/bin/env python3
"""One slab's kernels in launch order (for rocprofv3 --kernel-trace): N slabs of the headline domain through the
loopback transport, `steps` steps.  Post-processing (scripts/gpu_slab_ktrace.sh) prints the timeline of the last
step of the last slab: start offset, duration and the gap to the previous kernel's end.
  python scripts/studies/slab_step_trace.py [N] [steps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import cudafluidsimulator_amd as sph
from cudafluidsimulator_amd import mgpu as M

N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 12
mg = M.MultiGpuSimulator(sph.default_settings(4194304, True), world=N, transport="loopback")
mg.setup()
for _ in range(steps):
    mg.simulate()
mg.sync()
mg.close()
