"""Full-size checks (BASELINE.json configs 2 and 3) through size-independent
properties, plus bounded oracle comparisons at those sizes."""
import numpy as np
import pytest

import cudafluidsimulator_amd as sph
from cudafluidsimulator_amd import _lib
from helpers import assert_bit_equal
from oracle import oracle as O

pytestmark = pytest.mark.gpu


def run(n, steps, sweep="list", flags=0):
    sim = sph.Simulator(sph.default_settings(n, True), sweep=sweep, flags=flags)
    sim.setup()
    for _ in range(steps):
        sim.simulate()
    return sim


def test_config2_262144_all_sweeps_equal_and_oracle():
    a = run(262144, 12, "list", _lib.SPH_FLAG_COUNT_PAIRS)
    b = run(262144, 12, "direct")
    c = run(262144, 12, "lds")
    sa, sb, sc = a.download_state(), b.download_state(), c.download_state()
    for k in sa:
        assert_bit_equal(sa[k], sb[k], k)
        assert_bit_equal(sa[k], sc[k], k + " (lds)")
    c.close()
    ref = O.OracleSim(262144, True)
    ref.setup(); ref.step(12)
    assert_bit_equal(sa["pos"], ref.download()["pos"], "oracle")
    kt = a.kernel_times()
    assert kt.steps == 12 and kt.pair_tests > 12 * 3.5e6
    a.close(); b.close()


def test_config3_4194304_invariants_and_determinism():
    n, steps = 4194304, 4
    a = run(n, steps, flags=_lib.SPH_FLAG_COUNT_PAIRS)
    g = a.download_grid()
    assert np.array_equal(np.sort(g["ids"]), np.arange(n, dtype=np.uint32))   # permutation
    assert (np.diff(g["keys"].astype(np.int64)) >= 0).all()                    # sortedness
    cnt = g["cells"][:, 1] - g["cells"][:, 0]
    assert cnt.sum() == n and (cnt >= 0).all()                                 # cells partition
    st = a.download_state()
    h, hi = np.float32(0.1), np.float32(10.0) - np.float32(0.1)
    assert (st["pos"] >= h).all() and (st["pos"] <= hi).all()
    assert np.isfinite(st["vel"]).all() and (st["rho"] >= 31.3).all()
    assert_bit_equal(np.array(a.getPosition()), st["pos"], "getPosition order")
    kt = a.kernel_times()
    assert 8.5e8 * steps < kt.pair_tests < 9.6e8 * steps  # 9.09e8/sweep
    b = run(n, steps)                                      # run-to-run bit reproducible
    assert_bit_equal(b.download_state()["pos"], st["pos"], "determinism")
    b.close()
    # bounded oracle comparison at full size: 2 steps
    ref = O.OracleSim(n, True)
    ref.setup(); ref.step(2)
    c = run(n, 2)
    assert_bit_equal(c.download_state()["pos"], ref.download()["pos"], "oracle 2 steps @4M")
    a.close(); c.close()
