"""Pins the CPU oracle.  The reference ships NO tests, fixtures or golden data
(SURVEY.md section 4), so parity is unpinned by reference artefacts; what pins the
oracle are these known-answer values, hand-derived from the cited reference
formulas in strict fp32."""
import numpy as np
import pytest

from oracle import oracle as O


def hexf(x):
    return float(np.float32(x)).hex()


def test_constants_main_cpp_57_63():
    s = O.make_settings(1000, False)
    assert hexf(s.h) == "0x1.99999a0000000p-4"
    assert hexf(s.v_kernel_coeff) == "0x1.b521cc0000000p+23"   # 14323942
    assert hexf(s.d_kernel_coeff) == "0x1.7586a40000000p+30"   # 1.56668134e9
    assert hexf(s.timestep) == "0x1.47ae140000000p-7"
    assert s.boxDim == 10.0 and s.numCellsPerDim == 100.0
    assert hexf(np.float32(s.boxDim) - np.float32(s.h)) == "0x1.3cccccp+3".replace("p+3", "0000000p+3")


def test_single_particle_density_and_step():
    sim = O.OracleSim(1, False)
    sim.setup()
    sim.step()
    d = sim.download()
    assert float(d["rho"][0]) == 31.33363151550293
    assert float(d["prs"][0]) == 0.0
    # pre-clamp y = 0.09902 -> clamped to h, v.y bounced to +0.049
    assert float(d["pos"][0][1]) == 0.10000000149011612
    assert float(d["vel"][0][1]) == 0.04899999871850014
    assert float(d["pos"][0][0]) == 0.10000000149011612
    assert d["vel"][0][0] == 0 and d["vel"][0][2] == 0


def test_two_particles_same_cell():
    sim = O.OracleSim(2, False)
    sim.setup()
    sim.step()
    d = sim.download()
    assert [float(x) for x in d["rho"]] == [31.54854965209961, 31.54854965209961]


def test_lattice_interior_density():
    # 3x3x3... use a full small lattice: interior particle of a 5^3 block spaced 0.09
    g = (np.float32(0.1) + np.float32(0.09) * np.arange(5, dtype=np.float32)).astype(np.float32)
    x, y, z = np.meshgrid(g + np.float32(3), g + np.float32(3), g + np.float32(3), indexing="ij")
    pos = np.stack([x.ravel(), y.ravel(), z.ravel()], 1).astype(np.float32)
    s = O.make_settings(len(pos), False)
    keys = O.cell_keys(s, pos)
    perm = O.stable_sort(keys)
    cs, ce = O.cell_table(keys[perm])
    rho, prs = O.density(s, pos[perm], cs, ce)
    centre = np.where(perm == 62)[0][0]  # index (2,2,2)
    assert abs(float(rho[centre]) - 32.623138427734375) < 1e-4  # "approx" in Appendix D: lattice offset changes the last ulps
    assert prs.max() == 0


def test_grid_init_layout():
    s = O.make_settings(1295029, False)
    pos = O.init_positions(s)
    assert float(pos.max()) == pytest.approx(9.81999969, abs=1e-7)
    k = 109 * 109 * 3 + 109 * 5 + 7
    spacing = np.float32(0.9) * np.float32(0.1)
    assert float(spacing) == pytest.approx(0.0899999961, abs=1e-9)
    want = np.float32(0.1) + spacing * np.array([3, 5, 7], dtype=np.float32)
    assert np.array_equal(pos[k], want.astype(np.float32))


def test_random_init_first_particle():
    s = O.make_settings(4, True)
    pos = O.init_positions(s)
    assert ["%.9g" % v for v in pos[0]] == ["7.72150183", "4.15506363", "7.26479387"]
    # re-running gives the same stream (srand(1) == never-seeded state)
    assert np.array_equal(pos, O.init_positions(s))


def test_workload_statistics_match_survey_appendix_b():
    sim = O.OracleSim(8192, False)
    sim.setup()
    sim.step()
    assert sim.last_pair_tests() == pytest.approx(9.30e4, rel=0.01)
    sim.step(99)
    assert sim.last_pair_tests() == pytest.approx(7.64e5, rel=0.01)
    d = sim.download()
    assert float(d["rho"].max()) == pytest.approx(765, rel=0.01)
    assert (d["prs"] > 0).sum() == 0  # -i grid never activates pressure (F11)


def test_cell_hash_range_of_clamped_positions():
    s = O.make_settings(2, False)
    pos = np.array([[0.1, 0.1, 0.1], [9.9, 9.9, 9.9]], np.float32)
    pos[1] = np.float32(10.0) - np.float32(0.1)
    keys = O.cell_keys(s, pos)
    assert keys[0] == 1 + 100 + 10000
    assert keys[1] == 98 + 9800 + 980000
