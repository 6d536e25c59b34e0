"""HIP path vs the CPU oracle through the C-ABI (cudafluidsimulator_amd.Simulator
is a thin ctypes mirror of it).  Strict mode is BIT-EXACT: every comparison
below is on the raw fp32 words.  The oracle itself is pinned only by analytic
known answers -- the reference has no fixtures (parity unpinned, see
test_oracle_known_answers.py)."""
import os

import numpy as np
import pytest

import cudafluidsimulator_amd as sph
from cudafluidsimulator_amd import _lib
from helpers import assert_bit_equal, clustered_state, dense_block, random_state
from oracle import oracle as O

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
SWEEPS = ["list", "lds", "direct"]


def make_pair(n, random_init, sweep="list", flags=0, pos=None, vel=None):
    s = sph.default_settings(n, random_init)
    sim = sph.Simulator(s, sweep=sweep, flags=flags)
    ref = O.OracleSim(n, random_init)
    if pos is None:
        sim.setup()
        ref.setup()
    else:
        sim.upload_state(pos, vel)
        ref.upload(pos, vel)
    return sim, ref


def compare_state(sim, ref, what):
    g = sim.download_state()
    r = ref.download()
    for k in ("pos", "vel", "rho", "prs"):
        assert_bit_equal(g[k], r[k], f"{what}:{k}")
    assert_bit_equal(np.array(sim.getPosition()), r["pos"], f"{what}:getPosition")


def test_setup_initialisers_match_reference_rules():
    for rnd in (False, True):
        sim, ref = make_pair(5000, rnd)
        g = sim.download_state()
        assert_bit_equal(g["pos"], ref.download()["pos"], "init")
        assert not g["vel"].any()
        sim.close()


@pytest.mark.parametrize("sweep", SWEEPS)
def test_grid_phase_matches_oracle_sort_and_cell_table(sweep):
    pos, vel = clustered_state(30000, 5)
    sim, ref = make_pair(len(pos), False, sweep, pos=pos, vel=vel)
    sim.phase("grid")
    g = sim.download_grid()
    keys = O.cell_keys(ref.settings, pos)
    perm = O.stable_sort(keys)
    assert np.array_equal(g["ids"], perm)
    assert np.array_equal(g["keys"], keys[perm])
    cs, ce = O.cell_table(keys[perm])
    occ = ce > cs
    assert np.array_equal(g["cells"][occ, 0], cs[occ]) and np.array_equal(g["cells"][occ, 1], ce[occ])
    assert not g["cells"][~occ].any()
    sim.phase("density"); sim.phase("force"); sim.phase("readback")
    ref.step()
    compare_state(sim, ref, "after phases")
    sim.close()


@pytest.mark.parametrize("sweep", SWEEPS)
@pytest.mark.parametrize("case", ["grid8192", "random4096", "n1", "n2", "n130"])
def test_steps_bit_exact(case, sweep):
    n, rnd = {"grid8192": (8192, False), "random4096": (4096, True), "n1": (1, False),
              "n2": (2, False), "n130": (130, True)}[case]
    sim, ref = make_pair(n, rnd, sweep)
    for k in range(1, 11):
        sim.simulate()
        ref.step()
        if k in (1, 2, 10):
            compare_state(sim, ref, f"{case} step {k}")
    sim.close()


@pytest.mark.parametrize("sweep", SWEEPS)
def test_north_star_grid_100_steps(sweep):
    """BASELINE.json north star: positions after 100 steps on -i grid input.  The
    stated tolerance is 1e-5 relative; strict mode achieves exact equality."""
    sim, ref = make_pair(8192, False, sweep)
    times = sph.Times()
    for _ in range(100):
        sim.simulateAndTime(times)
    ref.step(100)
    got, want = np.array(sim.getPosition()), ref.download()["pos"]
    rel = np.abs(got - want) / np.maximum(np.abs(want), 1e-30)
    assert rel.max() <= 1e-5
    assert_bit_equal(got, want, "grid 100 steps")
    assert times.iters == 100 and times.sphUpdate > 0 and times.buildGrid > 0
    sim.close()


@pytest.mark.parametrize("sweep", SWEEPS)
def test_pressure_term_dense_block(sweep):
    """rho > REST_DENSITY: exercises pressureKernel (simulator.cu:99-117), which
    no reference initialiser reaches within 100 steps."""
    pos = dense_block(16)
    sim, ref = make_pair(len(pos), False, sweep, flags=_lib.SPH_FLAG_STORE_FORCE, pos=pos)
    sim.simulate(); ref.step()
    r = ref.download(want_force=True)
    assert (r["prs"] > 0).sum() > 1000
    assert_bit_equal(sim.download_force(), r["force"], "force step 1")
    compare_state(sim, ref, "dense step 1")
    for _ in range(19):
        sim.simulate()
    ref.step(19)
    compare_state(sim, ref, "dense step 20")
    sim.close()


@pytest.mark.parametrize("sweep", SWEEPS)
def test_skewed_occupancy_and_fast_particles(sweep):
    """Hundreds of particles per cell (chunked LDS window, FIFO drains) and wall
    bounces on every axis."""
    pos, vel = clustered_state(40000, 9)
    vel *= np.float32(20.0)
    sim, ref = make_pair(len(pos), False, sweep, flags=_lib.SPH_FLAG_STORE_FORCE, pos=pos, vel=vel)
    sim.simulate(); ref.step()
    assert_bit_equal(sim.download_force(), ref.download(want_force=True)["force"], "force")
    compare_state(sim, ref, "skew step 1")
    for _ in range(5):
        sim.simulate()
    ref.step(5)
    compare_state(sim, ref, "skew step 6")
    sim.close()


def test_coincident_particles_hit_eps_gates():
    """dist < EPS_F gates (simulator.cu:110,125) and identical positions."""
    base = np.array([[5.0, 5.0, 5.0]], np.float32)
    pos = np.repeat(base, 40, 0)
    pos[20:] += np.float32(5e-5)
    pos[30:] += np.float32(0.03)
    for sweep in SWEEPS:
        sim, ref = make_pair(len(pos), False, sweep, pos=pos)
        for _ in range(3):
            sim.simulate(); ref.step()
            compare_state(sim, ref, "coincident")
        sim.close()


def test_headline_config_matches_the_oracles_checksums():
    """BASELINE config 3 at FULL size (-n 4194304 -i random): sha256 of the raw fp32
    position and density arrays (particle-id order) after 1, 10 and 30 steps against
    the checksums the CPU oracle produced (tests/golden/make_golden.py --full)."""
    import hashlib
    import json
    gold = json.load(open(os.path.join(GOLD, "random4194304_sha256.json")))
    sim = sph.Simulator(sph.default_settings(gold["n"], True))
    sim.setup()
    done = 0
    for k in sorted(int(x) for x in gold["steps"]):
        for _ in range(k - done):
            sim.simulate()
        done = k
        st = sim.download_state()
        want = gold["steps"][str(k)]
        assert hashlib.sha256(np.ascontiguousarray(st["pos"]).tobytes()).hexdigest() == want["pos_sha256"], k
        assert hashlib.sha256(np.ascontiguousarray(st["rho"]).tobytes()).hexdigest() == want["rho_sha256"], k
        assert hashlib.sha256(np.ascontiguousarray(np.array(sim.getPosition())).tobytes()).hexdigest() == \
            want["pos_sha256"], k
    sim.close()


@pytest.mark.parametrize("name,checkpoints", [("grid2048", [1, 10, 100]), ("random4096", [1, 10, 100]),
                                              ("dense4096", [1, 5, 20])])
def test_committed_goldens(name, checkpoints):
    g = np.load(os.path.join(GOLD, name + ".npz"))
    if name == "dense4096":
        pos = dense_block(16)
        sim = sph.Simulator(sph.default_settings(len(pos), False))
        sim.upload_state(pos)
    else:
        n, rnd = (2048, False) if name == "grid2048" else (4096, True)
        sim = sph.Simulator(sph.default_settings(n, rnd))
        sim.setup()
    done = 0
    for k in checkpoints:
        for _ in range(k - done):
            sim.simulate()
        done = k
        assert_bit_equal(np.array(sim.getPosition()), g[f"pos_{k}"], f"{name} pos_{k}")
        if k == checkpoints[0]:
            st = sim.download_state()
            assert_bit_equal(st["rho"], g[f"rho_{k}"], f"{name} rho")
            assert_bit_equal(st["vel"], g[f"vel_{k}"], f"{name} vel")
    sim.close()


def test_click_impulse_matches_oracle():
    pos, vel = random_state(20000, 13, lo=1.0, hi=9.0, vmax=0.1)
    sim, ref = make_pair(len(pos), False, pos=pos, vel=vel)
    sim.simulate(); ref.step()
    sim.mouseClicked, sim.clickCoords = True, (420, 333)
    sim.simulate()            # step, then the click on that step's grid
    ref.step(); ref.click(420, 333)
    compare_state(sim, ref, "after click")
    sim.simulate(); ref.step()
    compare_state(sim, ref, "step after click")
    sim.close()


def test_fast_math_mode_within_north_star_tolerance():
    """SPH_MATH_FAST (FMA + approximate rcp/rsq) is NOT bit-exact; BASELINE.json's
    stated tolerance is 1e-5 relative on positions after 100 steps of -i grid."""
    n = 8192
    sim = sph.Simulator(sph.default_settings(n, False), math="fast")
    sim.setup()
    ref = O.OracleSim(n, False)
    ref.setup()
    for _ in range(100):
        sim.simulate()
    ref.step(100)
    got, want = np.array(sim.getPosition()), ref.download()["pos"]
    rel = np.abs(got - want) / np.maximum(np.abs(want), 1e-30)
    print("fast-math max relative position error after 100 grid steps:", rel.max())
    assert rel.max() <= 1e-5
    sim.close()
    # pressure-active dense block, 20 steps: looser sanity bound (chaotic growth)
    pos = dense_block(16)
    sim = sph.Simulator(sph.default_settings(len(pos), False), math="fast")
    sim.upload_state(pos)
    ref = O.OracleSim(len(pos), False)
    ref.upload(pos)
    sim.simulate(); ref.step()
    g, r = sim.download_state(), ref.download()
    assert np.allclose(g["rho"], r["rho"], rtol=1e-5)
    assert np.allclose(g["pos"], r["pos"], rtol=1e-5, atol=1e-7)
    sim.close()


@pytest.mark.parametrize("hh,box", [(0.2, 6.4), (0.25, 8.0)])
@pytest.mark.parametrize("sweep", SWEEPS)
def test_non_default_settings(sweep, hh, box):
    """Nothing is hard-wired to the reference's 100^3 grid: a 32-cell box with
    h = 0.2 or 0.25, a larger time step, 3001 particles.  h = 0.25f is a radius
    for which the largest dist2 with sqrtf(dist2) <= h lies one ulp ABOVE h*h
    (viscosityKernel tests r, pressureKernel r^2: simulator.cu:105,125), so the
    list sweep's general hit-bit path runs; for 0.1f and 0.2f the two cut-offs
    coincide and the sign-bit path runs."""
    import ctypes as C
    n = 3001
    rng = np.random.default_rng(3)
    s = sph.default_settings(n, False)
    s.h = hh
    s.boxDim = box
    s.numCellsPerDim = 32
    s.timestep = 0.004
    h = np.float32(s.h)
    s.v_kernel_coeff = float(np.float32(45.0) / (np.float32(3.14159265) * np.float32(float(h) ** 6)))
    s.d_kernel_coeff = float(np.float32(315.0) / (np.float32(64.0) * np.float32(3.14159265) * np.float32(float(h) ** 9)))
    pos = rng.uniform(1.5 * hh, box - 1.5 * hh, (n, 3)).astype(np.float32)
    vel = rng.uniform(-2, 2, (n, 3)).astype(np.float32)
    sim = sph.Simulator(s, sweep=sweep)
    sim.upload_state(pos, vel)
    ref = O.OracleSim(n, False)
    C.memmove(C.byref(ref.settings), C.byref(s), C.sizeof(s))
    ref.close()
    ref._h = O.lib().oracle_sim_create(C.byref(ref.settings))
    ref.upload(pos, vel)
    for k in range(6):
        sim.simulate(); ref.step()
    compare_state(sim, ref, "custom settings")
    sim.close()


def test_list_sweep_pool_exhaustion_falls_back(monkeypatch):
    """A mask pool that is too small must only cost speed: waves that find it
    exhausted mark their particles and the force sweep tests every candidate."""
    pos, vel = clustered_state(30000, 23)
    monkeypatch.setenv("SPH_MASK_POOL_WORDS", "20000")  # far too few words
    sim, ref = make_pair(len(pos), False, "list", pos=pos, vel=vel)
    monkeypatch.delenv("SPH_MASK_POOL_WORDS")
    for _ in range(3):
        sim.simulate(); ref.step()
    compare_state(sim, ref, "tiny pool")
    sim.close()


def test_checkpoint_resume_is_bit_identical(tmp_path):
    pos, vel = clustered_state(20000, 17)
    a, ref = make_pair(len(pos), False, pos=pos, vel=vel)
    for _ in range(4):
        a.simulate()
    snap = tmp_path / "state.sphsnap"
    a.save_state(snap)
    for _ in range(4):
        a.simulate()
    b = sph.Simulator(sph.default_settings(len(pos), False))
    b.load_state(snap)
    for _ in range(4):
        b.simulate()
    sa, sb = a.download_state(), b.download_state()
    for k in sa:
        assert_bit_equal(sa[k], sb[k], "resume:" + k)
    ref.step(8)
    assert_bit_equal(sb["pos"], ref.download()["pos"], "resume vs oracle")
    with pytest.raises(sph.SphError, match="match"):
        sph.Simulator(sph.default_settings(10, False)).load_state(snap)
    a.close(); b.close()


def test_api_errors():
    s = sph.default_settings(10, False)
    sim = sph.Simulator(s)
    with pytest.raises(sph.SphError):
        sim.simulate()  # before setup
    bad = np.full((10, 3), 11.0, np.float32)
    with pytest.raises(sph.SphError, match="outside"):
        sim.upload_state(bad)
    with pytest.raises(sph.SphError):
        sim.upload_state(np.zeros((5, 3), np.float32) + 1)
    sim.setup()
    with pytest.raises(sph.SphError):
        sim.moveParticles((400, 300))  # no step yet
    sim.close()
    sim0 = sph.Simulator(sph.default_settings(0, False))
    sim0.setup(); sim0.simulate(); sim0.simulate()
    assert sim0.getPosition().shape == (0, 3)
    sim0.close()


@pytest.fixture
def morton_oracle():
    """The oracle's key function is a process-wide switch: Morton inside the test only."""
    O.set_key_order("morton")
    yield
    O.set_key_order("flattened")


def _morton(cx, cy, cz):
    def spread(v):
        out = np.zeros_like(v)
        for b in range(7):
            out |= ((v >> b) & 1) << (3 * b)
        return out
    return spread(cx) | (spread(cy) << 1) | (spread(cz) << 2)


@pytest.mark.gpu
def test_morton_keys_match_the_oracle(morton_oracle):
    """SPH_KEY_MORTON (BASELINE config 3's ordering; SURVEY.md A.6: the oracle takes the key
    function as a parameter because it changes the tie order): grid phase and 12 steps of a
    clustered state against the Morton-keyed oracle, bit for bit; the cell table is indexed
    by the interleaved bits of the cell coordinates."""
    pos, vel = clustered_state(30000, 29)
    s = sph.default_settings(len(pos), False)
    sim = sph.Simulator(s, sweep="direct", key_order="morton")
    sim.upload_state(pos, vel)
    ref = O.OracleSim(len(pos), False)
    ref.upload(pos, vel)
    sim.phase("grid")
    g = sim.download_grid()
    c = (pos / np.float32(0.1)).astype(np.float32).astype(np.int64)
    keys = _morton(c[:, 0], c[:, 1], c[:, 2])
    order = np.argsort(keys, kind="stable")
    assert np.array_equal(g["ids"], order.astype(np.uint32))
    assert np.array_equal(g["keys"], keys[order].astype(np.uint32))
    assert len(g["cells"]) == 2 ** 21 == O.num_keys(100)
    sim.phase("density"); sim.phase("force"); sim.phase("readback")
    ref.step()
    compare_state(sim, ref, "morton step 1")
    for _ in range(11):
        sim.simulate(); ref.step()
    compare_state(sim, ref, "morton step 12")
    sim.close()
    ref.close()


@pytest.mark.gpu
def test_morton_and_flattened_orders_agree_to_rounding(morton_oracle):
    """The two key functions give different tie orders inside a cell, hence different fp32
    summation orders: results agree to rounding, not bit for bit (and both are legal
    outcomes of the reference's racy list order)."""
    pos = dense_block(14, jitter=0.003, seed=4)
    s = sph.default_settings(len(pos), False)
    out = {}
    for ko in ("flattened", "morton"):
        sim = sph.Simulator(s, sweep="direct", key_order=ko)
        sim.upload_state(pos)
        for _ in range(5):
            sim.simulate()
        out[ko] = sim.download_state()
        sim.close()
    assert np.allclose(out["morton"]["pos"], out["flattened"]["pos"], rtol=1e-5, atol=1e-6)
    assert (out["morton"]["rho"] > 1000).sum() > 100


@pytest.mark.gpu
def test_morton_needs_the_direct_sweep():
    with pytest.raises(sph.SphError, match="MORTON"):
        sph.Simulator(sph.default_settings(100, True), sweep="list", key_order="morton")


@pytest.mark.gpu
def test_corrupt_snapshot_is_rejected(tmp_path):
    """sph_load_state validates what it reads like sph_upload_state does: a NaN or an
    out-of-box position in the file must not reach the device."""
    pos, vel = random_state(3000, 41)
    sim = sph.Simulator(sph.default_settings(len(pos), False))
    sim.upload_state(pos, vel)
    sim.simulate()
    path = tmp_path / "snap.bin"
    sim.save_state(path)
    raw = bytearray(open(path, "rb").read())
    good = bytes(raw)
    for value in (np.float32("nan"), np.float32(123.0)):
        raw = bytearray(good)
        raw[64 + 16 * 7: 64 + 16 * 7 + 4] = np.float32(value).tobytes()   # x of row 7
        bad = tmp_path / "bad.bin"
        open(bad, "wb").write(bytes(raw))
        with pytest.raises(sph.SphError, match="corrupt snapshot"):
            sim.load_state(bad)
    sim.load_state(path)   # the intact file still loads
    sim.simulate()
    sim.close()


@pytest.mark.gpu
@pytest.mark.parametrize("n", [0, 1, 2, 63, 64, 65, 127, 129, 1000])
def test_odd_particle_counts(n):
    """Waves with 1..64 valid lanes, a launch of a single wave, no particle at all."""
    rng = np.random.default_rng(100 + n)
    pos = rng.uniform(4.0, 4.6, (n, 3)).astype(np.float32)   # dense enough to interact
    vel = rng.uniform(-1, 1, (n, 3)).astype(np.float32)
    for sweep in SWEEPS:
        sim, ref = make_pair(n, False, sweep, pos=pos, vel=vel)
        for _ in range(3):
            sim.simulate(); ref.step()
        if n:
            compare_state(sim, ref, f"n={n} {sweep}")
        sim.close()


# ---- zero-pair filter of the list force sweep (sweeps_list.hip: sl_is_quiet) ----

def co_moving_mixture(seed):
    """A cloud in which most particles share ONE velocity (quiet rows: their mutual pairs add
    exactly +-0 and are dropped unread) around (a) a block dense enough for pressure that moves
    with the SAME velocity (quiet velocity, but pressure: its pairs must be kept), (b) particles
    with velocities of their own scattered through the cloud, (c) a particle at rest."""
    rng = np.random.default_rng(seed)
    n = 40000
    pos = rng.uniform(2.0, 6.0, (n, 3)).astype(np.float32)
    vel = np.tile(np.array([0.25, -1.5, 0.125], np.float32), (n, 1))
    blk = dense_block(14, spacing=0.03, origin=(3.0, 3.0, 3.0), jitter=0.004, seed=seed)
    pos[:len(blk)] = blk                      # co-moving, under pressure
    odd = rng.choice(np.arange(len(blk), n), 3000, replace=False)
    vel[odd] = rng.uniform(-1, 1, (len(odd), 3)).astype(np.float32)
    vel[odd[0]] = 0.0
    # the last particle in sorted order sets the reference velocity: keep it a cloud particle
    pos[-1] = (5.99, 5.99, 5.99)
    return pos, vel


@pytest.mark.parametrize("seed", [1, 2])
def test_zero_pair_filter_mixed_quiet_and_active_rows(seed):
    pos, vel = co_moving_mixture(seed)
    sim, ref = make_pair(len(pos), False, "list", pos=pos, vel=vel)
    for k in range(1, 7):
        sim.simulate()
        ref.step()
        compare_state(sim, ref, f"co-moving mixture, step {k}")
    sim.close()


def test_zero_pair_filter_on_equals_off(monkeypatch):
    """SPH_ZERO_PAIR_FILTER=0 evaluates every recorded hit; the default drops the exact-zero
    pairs.  Same bits, and the counters show the filter actually dropped something."""
    pos, vel = co_moving_mixture(3)
    s = sph.default_settings(len(pos), False)
    out = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("SPH_ZERO_PAIR_FILTER", mode)
        sim = sph.Simulator(s, flags=_lib.SPH_FLAG_COUNT_PAIRS)
        sim.upload_state(pos, vel)
        for _ in range(4):
            sim.simulate()
        out[mode] = (sim.download_state(), sim.kernel_times().pair_hits, sim.debug_counters()[15])
        sim.close()
    for k in ("pos", "vel", "rho"):
        assert_bit_equal(out["1"][0][k], out["0"][0][k], f"filter on vs off: {k}")
    assert out["0"][1] == out["0"][2] == out["1"][2], "hits recorded do not depend on the filter"
    assert 0 < out["1"][1] < 0.6 * out["0"][1], "the filter dropped the co-moving cloud's pairs"


def test_zero_pair_filter_free_fall_drops_every_pair():
    """The reference's -i random start: everything at rest, no pressure -> every pair is an exact
    zero and the force sweep evaluates no pair body at all; results still equal the oracle's."""
    n = 100000
    s = sph.default_settings(n, True)
    sim = sph.Simulator(s, flags=_lib.SPH_FLAG_COUNT_PAIRS)
    ref = O.OracleSim(n, True)
    sim.setup(); ref.setup()
    for _ in range(3):
        sim.simulate(); ref.step()
    assert sim.kernel_times().pair_hits == 0 and sim.debug_counters()[15] > 0
    compare_state(sim, ref, "free fall")
    sim.close()


# ---- reachable product paths that had no test ----

@pytest.mark.parametrize("n", [4096, 262144])
def test_step_graphs_equal_the_oracle(n, monkeypatch):
    """SPH_GRAPH=1: the three phases of a step replayed as hipGraphs (captured once per
    read-back slot, sph_api.hip capture_step_graph).  Ten steps, bit-equal to the oracle after
    every one of them (getPosition() included: the read-back runs beside the replays), and the
    per-kernel event bookkeeping of the replays still adds up to ten steps."""
    monkeypatch.setenv("SPH_GRAPH", "1")
    sim, ref = make_pair(n, True)
    t = sph.Times()
    for k in range(1, 11):
        if k % 2:
            sim.simulate()
        else:
            sim.simulateAndTime(t)
        ref.step()
        assert_bit_equal(np.array(sim.getPosition()), ref.download()["pos"], f"graph replay, step {k}: getPosition")
    compare_state(sim, ref, "graph replay, step 10")
    kt = sim.kernel_times()
    assert kt.steps == 10 and kt.density > 0 and kt.force > 0 and t.iters == 5
    # a new state drops the captured graphs (their launches carry the old tile map)
    pos, vel = random_state(n, 4)
    sim.upload_state(pos, vel)
    ref.upload(pos, vel)
    for _ in range(3):
        sim.simulate()
        ref.step()
    compare_state(sim, ref, "graph replay after a re-upload")
    sim.close()


def test_click_after_graph_replays_walks_that_steps_grid(monkeypatch):
    """SPH_GRAPH=1 and the click impulse: the two read-back slots' graphs were captured with different
    cell tables (the grid build alternates between two), so after a REPLAY the click has to walk the
    table of the replayed slot, not the one the last capture left behind.  Clicks after steps 3, 4 and 7
    (both slots, all replays) against the oracle."""
    monkeypatch.setenv("SPH_GRAPH", "1")
    pos, vel = random_state(30000, 21, lo=1.0, hi=9.0, vmax=0.1)
    sim, ref = make_pair(len(pos), False, pos=pos, vel=vel)
    clicks = {3: (420, 333), 4: (250, 200), 7: (590, 440)}
    for step in range(1, 9):
        if step in clicks:
            sim.mouseClicked, sim.clickCoords = True, clicks[step]
        sim.simulate()
        ref.step()
        if step in clicks:
            ref.click(*clicks[step])
        compare_state(sim, ref, f"graph mode, step {step}")
    sim.close()


@pytest.mark.parametrize("sweep", ["list", "lds"])
def test_mapped_positions_after_every_step(sweep):
    """SPH_FLAG_MAPPED_POSITIONS (zero-copy display path): the force sweep
    scatters the id-ordered positions straight into host-mapped memory; getPosition() after
    each of five steps equals the oracle's positions, and so does the device state."""
    n = 50000
    s = sph.default_settings(n, True)
    sim = sph.Simulator(s, sweep=sweep, flags=_lib.SPH_FLAG_MAPPED_POSITIONS)
    ref = O.OracleSim(n, True)
    sim.setup(); ref.setup()
    for k in range(1, 6):
        sim.simulate()
        ref.step()
        assert_bit_equal(np.array(sim.getPosition()), ref.download()["pos"], f"mapped: getPosition after step {k}")
    compare_state(sim, ref, "mapped, step 5")
    sim.close()


# ---- the pair body's short divide / square-root chains are for the reference's constants only ----

def _custom_pair(hh, cells, n, pos, vel, sweep, dt=0.002):
    import ctypes as C
    s = sph.default_settings(n, False)
    s.h = hh
    s.boxDim = hh * cells
    s.numCellsPerDim = cells
    s.timestep = dt
    h = np.float32(s.h)
    s.v_kernel_coeff = float(np.float32(45.0) / (np.float32(3.14159265) * np.float32(float(h) ** 6)))
    s.d_kernel_coeff = float(np.float32(315.0) / (np.float32(64.0) * np.float32(3.14159265) * np.float32(float(h) ** 9)))
    sim = sph.Simulator(s, sweep=sweep)
    sim.upload_state(pos, vel)
    ref = O.OracleSim(n, False)
    C.memmove(C.byref(ref.settings), C.byref(s), C.sizeof(s))
    ref.close()
    ref._h = O.lib().oracle_sim_create(C.byref(ref.settings))
    ref.upload(pos, vel)
    return sim, ref


@pytest.mark.parametrize("hh", [0.01, 1.0])
@pytest.mark.parametrize("sweep", SWEEPS)
def test_extreme_but_legal_settings_use_the_full_ieee_pair_body(sweep, hh):
    """h = 0.01 and h = 1 (64-cell boxes): kernel coefficients 1e12 / 1e-6 times the reference's,
    densities from 1e-4 (the clamp) to 1e7, with a dense cluster, exactly coincident particles and
    pairs 2e-4 apart (just above the EPS gate).  Any h or coefficient other than the reference's
    selects the compiler's IEEE divide / square root in the pair body (DevParams::slimDiv): list,
    lds and direct sweeps all equal the oracle bit for bit."""
    rng = np.random.default_rng(11)
    cells, n = 64, 6000
    box = hh * cells
    pos = rng.uniform(1.5 * hh, box - 1.5 * hh, (n, 3)).astype(np.float32)
    pos[:1500] = (np.float32(box / 2) + rng.uniform(-0.6 * hh, 0.6 * hh, (1500, 3))).astype(np.float32)  # dense cluster
    pos[1500:1520] = pos[1520:1540]                                   # coincident pairs
    pos[1540:1560] = pos[1560:1580] + np.float32(2e-4)                # near-coincident pairs
    vel = rng.uniform(-20 * hh, 20 * hh, (n, 3)).astype(np.float32)
    sim, ref = _custom_pair(hh, cells, n, pos, vel, sweep)
    for k in range(1, 5):
        sim.simulate(); ref.step()
        compare_state(sim, ref, f"h={hh} {sweep} step {k}")
    sim.close()


def test_reference_settings_slim_equals_full_ieee(monkeypatch):
    """With the reference's constants the default pair body (bare Newton chains) and the full IEEE
    expansions (SPH_SLIM_DIV=0) give the same bits on a state with pressure, coincident and
    near-coincident particles; so does the check path (direct sweep: always the full expansions)."""
    pos = dense_block(20, spacing=0.02, jitter=0.004, seed=5)
    pos[10:20] = pos[30:40]
    pos[50:60] = pos[70:80] + np.float32(1.2e-4)
    vel = np.random.default_rng(6).uniform(-2, 2, pos.shape).astype(np.float32)
    s = sph.default_settings(len(pos), False)
    out = {}
    for name, env, sweep in (("slim", None, "list"), ("ieee", "0", "list"), ("direct", None, "direct")):
        if env is None:
            monkeypatch.delenv("SPH_SLIM_DIV", raising=False)
        else:
            monkeypatch.setenv("SPH_SLIM_DIV", env)
        sim = sph.Simulator(s, sweep=sweep)
        sim.upload_state(pos, vel)
        for _ in range(5):
            sim.simulate()
        out[name] = sim.download_state()
        sim.close()
    assert out["slim"]["rho"].max() > 2000
    for k in ("pos", "vel"):
        assert_bit_equal(out["slim"][k], out["ieee"][k], f"slim vs full IEEE: {k}")
        assert_bit_equal(out["slim"][k], out["direct"][k], f"list vs direct: {k}")


# ---- timed steps queue the next step's grid build before they wait (sph_handle::gridAhead) ----

def test_grid_built_ahead_survives_everything_that_can_happen_between_two_steps(tmp_path, monkeypatch):
    """simulateAndTime() queues the NEXT step's grid build before it waits for its own step.  Whatever
    comes between two steps must still see the state the finished step left, and the next step must be
    the oracle's: a click (walks the finished step's cell table, kept in a second table, and drops the
    grid built ahead -- it gathered velocities the impulse changes), getPosition(), download_state(), a
    snapshot, the phase API, a plain simulate(), a re-upload; and SPH_PIPELINE=0 gives the same bits."""
    pos, vel = random_state(30000, 31, lo=1.0, hi=9.0, vmax=0.3)

    def run(pipeline):
        monkeypatch.setenv("SPH_PIPELINE", pipeline)
        sim, ref = make_pair(len(pos), False, pos=pos, vel=vel)
        t = sph.Times()
        sim.simulateAndTime(t); ref.step()
        compare_state(sim, ref, "timed step 1")                       # state + getPosition with a grid ahead
        sim.simulateAndTime(t); ref.step()
        sim.moveParticles((420, 333)); ref.click(420, 333)            # click after a timed step
        sim.moveParticles((300, 250)); ref.click(300, 250)            # ... and a second one
        compare_state(sim, ref, "clicks after a timed step")
        sim.simulateAndTime(t); ref.step()
        snap = str(tmp_path / f"ahead{pipeline}.bin")
        sim.save_state(snap)                                          # snapshot with a grid ahead
        sim.simulate(); ref.step()                                    # untimed step consumes the grid ahead
        for name in ("grid", "density", "force", "readback"):         # the phase API after it
            sim.phase(name)
        ref.step()
        sim.simulateAndTime(t); ref.step()
        for name in ("grid", "density", "force", "readback"):         # the phase API right after a timed step
            sim.phase(name)
        ref.step()
        compare_state(sim, ref, "after phases")
        g = sim.download_grid()                                       # (no grid ahead now: this step's grid)
        assert np.array_equal(np.sort(g["ids"]), np.arange(len(pos), dtype=np.uint32))
        sim.simulateAndTime(t); ref.step()
        kt = sim.kernel_times()
        assert kt.steps == 6 and t.iters == 5   # (steps driven through the phase API carry no events)
        final = sim.download_state()
        compare_state(sim, ref, "end")
        # resume from the snapshot taken with a grid ahead: same continuation
        sim.load_state(snap)
        ref2 = O.OracleSim(len(pos), False)
        ref2.upload(pos, vel)
        for k in range(3):
            ref2.step()
            if k == 1:
                ref2.click(420, 333); ref2.click(300, 250)
        sim.simulateAndTime(t); ref2.step()
        compare_state(sim, ref2, "resumed")
        sim.upload_state(pos, vel)                                    # re-upload with a grid ahead
        ref3 = O.OracleSim(len(pos), False)
        ref3.upload(pos, vel)
        sim.simulateAndTime(t); ref3.step()
        compare_state(sim, ref3, "after a re-upload")
        sim.close()
        return final

    a, b = run("1"), run("0")   # (30,000 particles: on by default; "1" also covers handles above the size threshold)
    for k in ("pos", "vel", "rho"):
        assert_bit_equal(a[k], b[k], f"pipelined vs not: {k}")


@pytest.mark.parametrize("n", [100000, 2000000])
def test_timed_and_untimed_steps_mixed_read_back_paths(n):
    """Timed steps read their positions back through an SDMA engine (issued by the host once it has seen
    the force sweep finish), untimed ones through the stream-ordered HIP copy; both land in the one host
    buffer getPosition() returns.  Any mix of the two, with and without getPosition() in between, shows
    the oracle's positions after every step (n = 2,000,000: the copies are long enough to overlap)."""
    sim, ref = make_pair(n, True)
    t = sph.Times()
    pattern = "TTUUTUTTTUUUTTUT"
    for k, c in enumerate(pattern, 1):
        if c == "T":
            sim.simulateAndTime(t)
        else:
            sim.simulate()
        ref.step()
        if k % 3 != 0:   # (every third step nobody looks: the next copy must still wait for this one)
            assert_bit_equal(np.array(sim.getPosition()), ref.download()["pos"], f"step {k} ({c})")
    compare_state(sim, ref, "end of the mixed run")
    assert t.iters == pattern.count("T") and sim.kernel_times().readback > 0
    sim.close()
