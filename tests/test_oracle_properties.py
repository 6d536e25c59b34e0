"""Invariants of the oracle step and committed goldens."""
import os

import numpy as np
import pytest

from helpers import assert_bit_equal, clustered_state, dense_block, random_state
from oracle import oracle as O

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_sort_is_stable_permutation_and_cells_partition():
    pos, _ = clustered_state(20000, 3)
    s = O.make_settings(len(pos), False)
    keys = O.cell_keys(s, pos)
    perm = O.stable_sort(keys)
    assert np.array_equal(np.sort(perm), np.arange(len(pos)))
    assert np.array_equal(perm, np.argsort(keys, kind="stable"))
    cs, ce = O.cell_table(keys[perm])
    cnt = ce - cs
    assert cnt.sum() == len(pos) and (cnt >= 0).all()
    occ = np.nonzero(cnt)[0]
    assert np.array_equal(cs[occ][1:], ce[occ][:-1])  # ranges tile the array


def test_positions_stay_in_box_and_ids_permute():
    pos, vel = random_state(5000, 7, vmax=30.0)  # fast particles hit the walls
    sim = O.OracleSim(len(pos), False)
    sim.upload(pos, vel)
    sim.step(20)
    d = sim.download()
    h, hi = np.float32(0.1), np.float32(10.0) - np.float32(0.1)
    assert (d["pos"] >= h).all() and (d["pos"] <= hi).all()
    assert np.array_equal(np.sort(sim.sorted_state()["ids"]), np.arange(len(pos)))
    assert np.isfinite(d["vel"]).all()


def test_dense_block_activates_pressure():
    pos = dense_block(16)
    sim = O.OracleSim(len(pos), False)
    sim.upload(pos)
    sim.step()
    d = sim.download()
    assert d["rho"].max() > 1000 and (d["prs"] > 0).sum() > 1000


def test_result_independent_of_thread_count():
    import subprocess, sys
    code = ("import sys; sys.path.insert(0, %r); import numpy as np; from oracle import oracle as O;"
            "s=O.OracleSim(3000, True); s.setup(); s.step(5); "
            "print(s.download()['pos'].view(np.uint32).sum(dtype=np.uint64))" %
            os.path.dirname(os.path.dirname(__file__)))
    outs = []
    for t in ("1", "4"):
        env = dict(os.environ, OMP_NUM_THREADS=t)
        outs.append(subprocess.check_output([sys.executable, "-c", code], env=env).strip())
    assert outs[0] == outs[1]


def test_baseline_build_of_the_oracle_gives_identical_bits():
    """bench.py times the -O3 -march=native build (oracle/Makefile `native`) as the CPU
    baseline; same source, still no contraction / fast-math, so the same results."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(__file__))
    code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r); import numpy as np; import hashlib;"
            "from oracle import oracle as O; from helpers import dense_block;"
            "NATIVE and O.use_native_build();"
            "p=dense_block(14, jitter=0.003, seed=2); s=O.OracleSim(len(p), False); s.upload(p); s.step(6);"
            "d=s.download(); print(hashlib.sha256(d['pos'].tobytes()+d['rho'].tobytes()).hexdigest())" %
            (root, os.path.dirname(__file__)))
    outs = [subprocess.check_output([sys.executable, "-c", code.replace("NATIVE", flag)]).strip()
            for flag in ("False", "True")]
    assert outs[0] == outs[1]


def test_click_impulse_pushes_column():
    pos, vel = random_state(4000, 11, vmax=0.0)
    sim = O.OracleSim(len(pos), False)
    sim.upload(pos, vel)
    sim.step()
    before = sim.download()["vel"].copy()
    sim.click(400, 300)  # centre of the click box -> cell (50, 100-50, *)
    after = sim.download()["vel"]
    changed = np.any(before != after, axis=1)
    assert 0 < changed.sum() < len(pos)


@pytest.mark.parametrize("name", ["grid2048", "random4096", "dense4096"])
def test_oracle_reproduces_committed_goldens(name):
    import sys
    sys.path.insert(0, GOLD)
    import make_golden
    path = os.path.join(GOLD, name + ".npz")
    assert os.path.exists(path), "run tests/golden/make_golden.py"
    g = np.load(path)
    out = make_golden.CASES[name]()
    for k in g.files:
        assert_bit_equal(out[k], g[k], f"{name}:{k}")


def test_oracle_reproduces_the_full_size_checksum_at_step_1():
    """The n = 4,194,304 sha256 fixture is an oracle output: re-derive its first entry
    (one step; the later entries cost minutes and are left to make_golden.py --full)."""
    import hashlib
    import json
    gold = json.load(open(os.path.join(GOLD, "random4194304_sha256.json")))
    sim = O.OracleSim(gold["n"], True)
    sim.setup()
    sim.step(1)
    d = sim.download()
    assert hashlib.sha256(np.ascontiguousarray(d["pos"]).tobytes()).hexdigest() == gold["steps"]["1"]["pos_sha256"]
    assert hashlib.sha256(np.ascontiguousarray(d["rho"]).tobytes()).hexdigest() == gold["steps"]["1"]["rho_sha256"]
    sim.close()
