"""Random sequences of the C-ABI's calls under random library knobs, against the oracle after every step.
Each seed draws the sweep family, the handle flags, the environment knobs that change the step's control flow
(graph replay, the grid build queued ahead, the read-back path, the zero-pair filter, the pair body's divide
chains) and ~30 operations: simulate / simulateAndTime / the four phases by hand / click / getPosition /
upload of a new state / setup() / save + load of a snapshot.  Strict mode: every comparison is bit-exact."""
import numpy as np
import pytest

import cudafluidsimulator_amd as sph
from cudafluidsimulator_amd import _lib
from helpers import assert_bit_equal, clustered_state, random_state
from oracle import oracle as O

pytestmark = pytest.mark.gpu
KNOBS = {"SPH_GRAPH": ("0", "1"), "SPH_PIPELINE": ("0", "1"), "SPH_READBACK_SDMA": ("0", "1"),
         "SPH_ZERO_PAIR_FILTER": ("0", "1"), "SPH_SLIM_DIV": ("0", "1"), "SPH_XCD_ROTATE": ("0", "1", "5")}


def check(sim, ref, what):
    g, r = sim.download_state(), ref.download()
    for k in ("pos", "vel", "rho"):
        assert_bit_equal(g[k], r[k], f"{what}: {k}")
    assert_bit_equal(np.array(sim.getPosition()), r["pos"], f"{what}: getPosition()")


@pytest.mark.parametrize("seed", list(range(40)) + [100, 101, 102, 200])
def test_random_call_sequences_stay_on_the_oracle(seed, monkeypatch, tmp_path):
    """seeds >= 100: n = 262,144 (several sort tiles, the grid build queued ahead by default), seed 200:
    n = 1,600,000 (past the size at which the sort switches to 4096-key tiles and pipelining goes off)."""
    rng = np.random.default_rng(1000 + seed)
    sweep = ["list", "list", "lds", "direct"][rng.integers(4)]
    flags = int(rng.choice([0, 0, _lib.SPH_FLAG_MAPPED_POSITIONS])) if sweep != "direct" else 0
    env = {k: str(rng.choice(v)) for k, v in KNOBS.items()}
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    n = int(rng.choice([1500, 6000, 20000]))
    n_ops = 30
    if seed >= 100:
        n, n_ops = (1600000, 8) if seed >= 200 else (262144, 14)
    random_init = bool(rng.integers(2))
    sim = sph.Simulator(sph.default_settings(n, random_init), sweep=sweep, flags=flags)
    ref = O.OracleSim(n, random_init)
    sim.setup()
    ref.setup()
    what = f"seed {seed} ({sweep}, flags {flags}, {env}, n {n})"
    t = sph.Times()
    steps = 0
    for op_no in range(n_ops):
        op = rng.choice(["step", "step", "step", "timed", "timed", "phases", "click", "peek", "upload", "setup", "snapshot",
                         "misuse"])
        tag = f"{what}, op {op_no} {op}"
        if op == "misuse":
            # calls that must fail loudly and leave the simulator as it was
            kind = rng.integers(5)
            with pytest.raises(sph.SphError):
                if kind == 0:
                    sim.phase(["force", "readback"][rng.integers(2)])                  # phases out of order
                elif kind == 1:
                    sim.upload_state(np.zeros((n + 1, 3), np.float32) + 5.0)           # wrong particle count
                elif kind == 2:
                    bad = np.full((n, 3), 5.0, np.float32); bad[n // 2, 1] = np.nan
                    sim.upload_state(bad)                                              # NaN
                elif kind == 3:
                    bad = np.full((n, 3), 5.0, np.float32); bad[0, 2] = 10.5
                    sim.upload_state(bad)                                              # outside the box
                else:
                    sim.load_state(tmp_path / "no_such_file.sphsnap")
        elif op == "step":
            sim.simulate(); ref.step(); steps += 1
        elif op == "timed":
            sim.simulateAndTime(t); ref.step(); steps += 1
        elif op == "phases":
            for ph in ("grid", "density", "force", "readback"):
                sim.phase(ph)
            ref.step(); steps += 1
        elif op == "click":
            xy = (int(rng.integers(100, 700)), int(rng.integers(100, 500)))
            sim.mouseClicked, sim.clickCoords = True, xy
            sim.simulate(); ref.step(); ref.click(*xy); steps += 1
        elif op == "peek":
            assert_bit_equal(np.array(sim.getPosition()), ref.download()["pos"], tag)
            continue
        elif op == "upload":
            pos, vel = (clustered_state if rng.integers(2) else random_state)(n, int(rng.integers(1 << 30)))
            sim.upload_state(pos, vel); ref.upload(pos, vel)
        elif op == "setup":
            sim.setup(); ref.setup()
        elif op == "snapshot":
            path = tmp_path / f"s{op_no}.sphsnap"
            sim.save_state(path)
            sim.simulate()                       # (moves on, then comes back to the snapshot)
            sim.load_state(path)
        check(sim, ref, tag)
    sim.close()


@pytest.mark.parametrize("seed", range(28))
def test_random_call_sequences_over_slabs_stay_on_the_oracle(seed, monkeypatch):
    """The same for the multi-GPU driver (all slabs on this GPU): world size, transport, re-cut period and the
    per-slab knobs drawn at random; steps, timed steps, clicks, getPosition(), uploads and setup() against the
    oracle after every operation."""
    from cudafluidsimulator_amd import mgpu as M
    rng = np.random.default_rng(5000 + seed)
    env = {k: str(rng.choice(v)) for k, v in KNOBS.items() if k in ("SPH_ZERO_PAIR_FILTER", "SPH_SLIM_DIV", "SPH_XCD_ROTATE")}
    env["SPH_MGPU_THREADS"] = str(rng.integers(2))
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    world = int(rng.integers(2, 7))
    transport = ["loopback", "streams", "rccl_self"][rng.integers(3)] if world <= 3 else ["loopback", "streams"][rng.integers(2)]
    recut = int(rng.choice([0, 0, 2, 5]))
    n = int(rng.choice([6000, 20000, 60000] if seed < 16 else [300, 2000, 6000]))  # (seeds >= 16: nearly empty slabs)
    random_init = bool(rng.integers(2))
    mg = M.MultiGpuSimulator(sph.default_settings(n, random_init), world=world, transport=transport, recut_every=recut)
    ref = O.OracleSim(n, random_init)
    mg.setup()
    ref.setup()
    what = f"seed {seed} ({world} slabs, {transport}, recut {recut}, {env}, n {n})"
    t = sph.Times()
    for op_no in range(24):
        op = rng.choice(["step", "step", "step", "timed", "click", "peek", "upload", "setup"])
        tag = f"{what}, op {op_no} {op}"
        if op == "step":
            mg.simulate(); ref.step()
        elif op == "timed":
            mg.simulateAndTime(t); ref.step()
        elif op == "click":
            xy = (int(rng.integers(100, 700)), int(rng.integers(100, 500)))
            mg.mouseClicked, mg.clickCoords = True, xy
            mg.simulate(); ref.step(); ref.click(*xy)
        elif op == "peek":
            assert_bit_equal(np.array(mg.getPosition()), ref.download()["pos"], tag)
            continue
        elif op == "upload":
            # (a moving cloud inside the box: z-layers 10..89 occupied, like the reference's random fill)
            pos, vel = random_state(n, int(rng.integers(1 << 30)), lo=1.0, hi=9.0, vmax=0.5)
            mg.upload_state(pos, vel); ref.upload(pos, vel)
        elif op == "setup":
            mg.setup(); ref.setup()
        g, r = mg.download_state(), ref.download()
        assert g["written"] == n
        for k in ("pos", "vel", "rho"):
            assert_bit_equal(g[k], r[k], f"{tag}: {k}")
        assert_bit_equal(np.array(mg.getPosition()), r["pos"], f"{tag}: getPosition()")
    mg.close()


def nasty_state(n, seed, fast_factor=30.0):
    """Positions on the box faces and in the corner cells, on exact multiples of the cell size, coincident,
    several hundred in one cell, a thin sheet; velocities up to 30 (three cells per step)."""
    rng = np.random.default_rng(seed)
    pos = rng.uniform(0.1, 9.9, (n, 3)).astype(np.float32)
    kind = rng.integers(0, 8, n)
    face = kind == 0
    pos[face] = np.where(rng.random((int(face.sum()), 3)) < 0.5, np.float32(0.1), np.float32(9.9))
    onh = kind == 1                                              # exact multiples of h = 0.1f
    pos[onh] = (rng.integers(1, 99, (int(onh.sum()), 3)).astype(np.float32) * np.float32(0.1))
    same = kind == 2                                             # coincident (dist = 0: every term gated out)
    pos[same] = np.float32(5.0)
    crowd = kind == 3                                            # one crowded cell and its neighbours
    pos[crowd] = (3.0 + 0.25 * rng.random((int(crowd.sum()), 3))).astype(np.float32)
    sheet = kind == 4                                            # a sheet one cell thick
    pos[sheet, 1] = (0.1 + 0.09 * rng.random(int(sheet.sum()))).astype(np.float32)
    pos = np.clip(pos, np.float32(0.1), np.float32(9.9))
    vel = rng.uniform(-1.0, 1.0, (n, 3)).astype(np.float32)
    fast = rng.random(n) < 0.02
    vel[fast] *= np.float32(fast_factor)
    return pos, vel


@pytest.mark.parametrize("seed", range(12))
def test_nasty_states_stay_on_the_oracle(seed, monkeypatch):
    rng = np.random.default_rng(9000 + seed)
    if seed % 3 == 0:  # a hit-stream pool far too small: some waves record, the rest go through k_force_fallback
        monkeypatch.setenv("SPH_MASK_POOL_WORDS", str(int(rng.choice([4096, 60000, 400000]))))
    n = int(rng.choice([3000, 12000, 40000]))
    sweep = ["list", "list", "lds", "direct"][rng.integers(4)]
    pos, vel = nasty_state(n, seed)
    sim = sph.Simulator(sph.default_settings(n, False), sweep=sweep)
    ref = O.OracleSim(n, False)
    sim.upload_state(pos, vel)
    ref.upload(pos, vel)
    for k in range(4):
        sim.simulate(); ref.step()
        check(sim, ref, f"nasty seed {seed} ({sweep}, n {n}), step {k + 1}")
    sim.close()


@pytest.mark.parametrize("seed", range(8))
def test_nasty_states_over_slabs_stay_on_the_oracle(seed, monkeypatch):
    """The same states cut into z-slabs (fast particles cross one or two z-layers per step: migration through the
    exchange; faces, coincident points and the crowded cell land on slab boundaries for some cuts)."""
    from cudafluidsimulator_amd import mgpu as M
    rng = np.random.default_rng(9500 + seed)
    if seed % 2 == 0:  # (per slab: interior and boundary launches both walk the list of waves without a stream)
        monkeypatch.setenv("SPH_MASK_POOL_WORDS", str(int(rng.choice([4096, 60000]))))
    n = int(rng.choice([3000, 12000, 40000]))
    world = int(rng.integers(2, 7))
    transport = ["loopback", "streams"][rng.integers(2)]
    pos, vel = nasty_state(n, 50 + seed, fast_factor=15.0)
    mg = M.MultiGpuSimulator(sph.default_settings(n, False), world=world, transport=transport, recut_every=int(rng.choice([0, 2])))
    ref = O.OracleSim(n, False)
    mg.upload_state(pos, vel)
    ref.upload(pos, vel)
    for k in range(4):
        mg.simulate(); ref.step()
        g, r = mg.download_state(), ref.download()
        for key in ("pos", "vel", "rho"):
            assert_bit_equal(g[key], r[key], f"nasty seed {seed} ({world} slabs, {transport}, n {n}), step {k + 1}: {key}")
    assert_bit_equal(np.array(mg.getPosition()), ref.download()["pos"], "getPosition()")
    mg.close()
