"""Generates tests/golden/*.npz from the CPU oracle (oracle/sph_oracle.c).

The reference holds no golden vectors for this path and its CUDA
source cannot run here, so these fixtures are ORACLE outputs: they pin the HIP
path and the oracle to each other across rounds, not to the reference.  Data
only: inputs are regenerated from seeds / the reference initialisers.
Run:  python tests/golden/make_golden.py          (the small array fixtures)
      python tests/golden/make_golden.py --full   (sha256 of the n = 4,194,304 run)
      python tests/golden/make_golden.py --config4 | --config5   (first steps of configs 4 / 5 at full size)
      python tests/golden/make_golden.py --morton  (Morton-keyed oracle, n = 4,194,304, moving state, steps 1 and 3)
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from helpers import dense_block  # noqa: E402
from oracle import oracle as O  # noqa: E402


def _run(sim, checkpoints):
    out = {}
    done = 0
    for k in checkpoints:
        sim.step(k - done)
        done = k
        d = sim.download()
        out[f"pos_{k}"] = d["pos"]
        if k == checkpoints[0]:
            out[f"rho_{k}"] = d["rho"]
            out[f"vel_{k}"] = d["vel"]
    return out


def grid2048():
    sim = O.OracleSim(2048, False)
    sim.setup()
    return _run(sim, [1, 10, 100])


def random4096():
    sim = O.OracleSim(4096, True)
    sim.setup()
    return _run(sim, [1, 10, 100])


def dense4096():
    pos = dense_block(16)  # rho ~ 1.3e3 > REST_DENSITY: pressure term active
    sim = O.OracleSim(len(pos), False)
    sim.upload(pos)
    return _run(sim, [1, 5, 20])


CASES = {"grid2048": grid2048, "random4096": random4096, "dense4096": dense4096}


def full_size_checksums(n=4194304, checkpoints=(1, 10, 30, 50, 60, 80, 100), dump_dir=None):
    """BASELINE config 3 (-n 4194304 -i random) is too big to commit as arrays: the
    fixture is the sha256 of the oracle's position / density arrays (particle-id order,
    raw fp32 bytes) at steps through the whole 100-step run: pressure switches on at
    step ~50 and cell occupancy reaches ~144 per cell by step 100,
    so the late checkpoints cover pressureKernel's 1/r branch and the dense-cell paths
    at full size.  About ten minutes of CPU on 8 cores.  dump_dir: also save the
    oracle's key-sorted positions at each checkpoint (workload studies; not committed)."""
    import hashlib
    import json
    sim = O.OracleSim(n, True)
    sim.setup()
    out = {"n": n, "init": "random", "order": "particle id", "dtype": "<f4", "steps": {}}
    done = 0
    for k in checkpoints:
        sim.step(k - done)
        done = k
        d = sim.download()
        out["steps"][str(k)] = {"pos_sha256": hashlib.sha256(np.ascontiguousarray(d["pos"]).tobytes()).hexdigest(),
                                "rho_sha256": hashlib.sha256(np.ascontiguousarray(d["rho"]).tobytes()).hexdigest()}
        out["steps"][str(k)]["pair_tests"] = int(sim.last_pair_tests())
        out["steps"][str(k)]["rho_max"] = float(d["rho"].max())
        out["steps"][str(k)]["particles_with_pressure"] = int((d["rho"] > 1000.0).sum())
        print(n, k, out["steps"][str(k)], flush=True)
        if dump_dir:
            srt = sim.sorted_state()
            np.savez(os.path.join(dump_dir, f"sorted_{n}_{k}.npz"), keys=srt["keys"], pos=srt["pos"])
    with open(os.path.join(HERE, f"random{n}_sha256.json"), "w") as f:
        json.dump(out, f, indent=1)

def morton_state(n=4194304, seed=20):
    """Input of the Morton fixture: the reference's -i random positions with seeded random velocities
    (up to 0.3 cells per step along every axis).  The reference's own start -- everything at rest --
    falls as one body for 45 steps: cells only exchange particles along y, and then the Morton and
    the flattened key put every cell's particles in the SAME order; with migration along all three
    axes they do not."""
    sim = O.OracleSim(n, True)
    sim.setup()
    pos = sim.download()["pos"].copy()
    sim.close()
    vel = np.random.default_rng(seed).uniform(-3.0, 3.0, (n, 3)).astype(np.float32)
    return pos, vel


def morton_checksums(n=4194304, checkpoints=(1, 3)):
    """BASELINE config 3 names the Morton ordering: the oracle with its key function switched to
    Morton (oracle_set_key_order) on morton_state(), sha256 after steps 1 and 3."""
    import hashlib
    import json
    pos, vel = morton_state(n)
    out = {"n": n, "init": "random positions + seeded velocities (make_golden.morton_state)", "seed": 20,
           "order": "particle id", "dtype": "<f4", "steps": {}, "flattened_steps": {}}
    for ko in ("morton", "flattened"):
        O.set_key_order(ko)
        sim = O.OracleSim(n, True)
        sim.upload(pos, vel)
        done = 0
        for k in checkpoints:
            sim.step(k - done)
            done = k
            d = sim.download()
            out["steps" if ko == "morton" else "flattened_steps"][str(k)] = {
                "pos_sha256": hashlib.sha256(np.ascontiguousarray(d["pos"]).tobytes()).hexdigest(),
                "rho_sha256": hashlib.sha256(np.ascontiguousarray(d["rho"]).tobytes()).hexdigest()}
            print(ko, n, k, flush=True)
        sim.close()
    O.set_key_order("flattened")
    with open(os.path.join(HERE, f"random{n}_morton_sha256.json"), "w") as f:
        json.dump(out, f, indent=1)


def big_config_checksums(n, random_init, checkpoints):
    """BASELINE configs 4 (-n 16777216 -i random) and 5 (-n 67108864 -i grid, the dense-lattice
    extension): sha256 of the oracle's arrays after the first step(s) at FULL size.  Config 5 is
    ~1.3e11 pair tests per sweep: a few CPU-minutes per step."""
    import hashlib
    import json
    sim = O.OracleSim(n, random_init)
    sim.setup()
    init = "random" if random_init else "grid"
    out = {"n": n, "init": init, "order": "particle id", "dtype": "<f4", "steps": {}}
    done = 0
    for k in checkpoints:
        sim.step(k - done)
        done = k
        d = sim.download()
        out["steps"][str(k)] = {"pos_sha256": hashlib.sha256(np.ascontiguousarray(d["pos"]).tobytes()).hexdigest(),
                                "rho_sha256": hashlib.sha256(np.ascontiguousarray(d["rho"]).tobytes()).hexdigest(),
                                "pair_tests": int(sim.last_pair_tests()), "rho_max": float(d["rho"].max()),
                                "particles_with_pressure": int((d["rho"] > 1000.0).sum())}
        print(n, k, out["steps"][str(k)], flush=True)
    with open(os.path.join(HERE, f"{init}{n}_sha256.json"), "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    if "--morton" in sys.argv:
        morton_checksums()
        sys.exit(0)
    if "--config4" in sys.argv:
        big_config_checksums(16777216, True, (1, 2))
        sys.exit(0)
    if "--config5" in sys.argv:
        big_config_checksums(67108864, False, (1,))
        sys.exit(0)
    if "--full" in sys.argv:
        dd = sys.argv[sys.argv.index("--dump-dir") + 1] if "--dump-dir" in sys.argv else None
        full_size_checksums(dump_dir=dd)
        sys.exit(0)
    for name, fn in CASES.items():
        out = fn()
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
        print(name, {k: v.shape for k, v in out.items()})
