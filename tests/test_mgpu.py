"""The in-process C++ multi-GPU driver (libsph_mgpu.so, include/sph_mgpu.h): N z-slabs must
reproduce the single-domain result BIT FOR BIT.  A one-GPU box can only run the loopback
transport (N slabs on one device, device-to-device copies) and RCCL with one rank sending
to itself; both go through the same step logic as the real multi-GPU transport."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

import cudafluidsimulator_amd as sph
from cudafluidsimulator_amd import mgpu as M
from helpers import assert_bit_equal

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "sph_mgpu.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(sph_mgpu_[a-z_]+)\s*\(", text)))


def test_mgpu_library_exports_every_declared_symbol():
    assert declared_symbols() == sorted(M.EXPORTED_SYMBOLS)
    L = M.load_mgpu_library()
    for name in declared_symbols():
        assert hasattr(L, name), name
    out = subprocess.check_output(["nm", "-D", "--defined-only", M.library_path()], text=True)
    assert set(declared_symbols()) <= set(re.findall(r" T (sph_mgpu_\w+)", out))
    assert C.sizeof(M.SphMgpuOptions) == 4 * (5 + 8 + 5)
    # the host logic links RCCL and the kernel library, never the oracle
    deps = subprocess.check_output(["ldd", M.library_path()], text=True)
    assert "librccl" in deps and "libsph_hip" in deps and "oracle" not in deps


def test_mgpu_without_a_gpu_fails_loudly(have_gpu):
    if have_gpu:
        pytest.skip("GPU present")
    with pytest.raises(sph.SphError, match="no HIP device"):
        M.MultiGpuSimulator(sph.default_settings(64, False), world=2, transport="loopback")


def moving_state(n, seed, vz=9.0):
    """Random fluid whose z-velocities move particles between slabs every step."""
    rng = np.random.default_rng(seed)
    pos = rng.uniform(0.5, 9.5, (n, 3)).astype(np.float32)
    vel = rng.uniform(-1, 1, (n, 3)).astype(np.float32)
    vel[:, 2] = rng.uniform(-vz, vz, n).astype(np.float32)
    return pos, vel


def single_domain(settings, pos, vel, steps, sweep="list"):
    sim = sph.Simulator(settings, sweep=sweep)
    if pos is None:
        sim.setup()
    else:
        sim.upload_state(pos, vel)
    for _ in range(steps):
        sim.simulate()
    st = sim.download_state()
    got = np.array(sim.getPosition(), copy=True)
    sim.close()
    return st, got


@pytest.mark.gpu
@pytest.mark.parametrize("world,transport", [(2, "loopback"), (4, "loopback"), (8, "loopback"), (3, "rccl_self"),
                                             (2, "streams"), (5, "streams"), (8, "streams")])
def test_slabs_equal_single_domain(world, transport):
    """"streams": the RCCL transport's stream layout (per-slab compute / exchange / boundary
    streams, exchange B beside the interior force sweep) with event-ordered copies as messages."""
    n, steps = 80000, 8
    pos, vel = moving_state(n, 5)
    settings = sph.default_settings(n, False)
    want, want_pos = single_domain(settings, pos, vel, steps)
    mg = M.MultiGpuSimulator(settings, world=world, transport=transport)
    mg.upload_state(pos, vel)
    for _ in range(steps):
        mg.simulate()
    got = mg.download_state()
    assert got["written"] == n
    assert_bit_equal(got["pos"], want["pos"], "pos")
    assert_bit_equal(got["vel"], want["vel"], "vel")
    assert_bit_equal(got["rho"], want["rho"], "rho")
    assert_bit_equal(np.array(mg.getPosition()), want_pos, "getPosition()")
    st = mg.stats()
    assert st.steps == steps and st.host_syncs == steps, "one host synchronisation per step"
    assert sum(st.owned[:world]) == n
    mg.close()


@pytest.mark.gpu
@pytest.mark.parametrize("world,transport", [(4, "loopback"), (8, "streams")])
def test_slab_worker_threads_equal_single_domain(world, transport, monkeypatch):
    """SPH_MGPU_THREADS=1: the per-slab parts of a step run on one host thread per slab (message
    rounds and the host synchronisation stay on the calling thread).  Same bits."""
    monkeypatch.setenv("SPH_MGPU_THREADS", "1")
    n, steps = 80000, 8
    pos, vel = moving_state(n, 5)
    settings = sph.default_settings(n, False)
    want, want_pos = single_domain(settings, pos, vel, steps)
    mg = M.MultiGpuSimulator(settings, world=world, transport=transport, recut_every=3)
    mg.upload_state(pos, vel)
    t = sph.Times()
    for _ in range(steps):
        mg.simulateAndTime(t)
    got = mg.download_state()
    assert_bit_equal(got["pos"], want["pos"], "pos")
    assert_bit_equal(got["rho"], want["rho"], "rho")
    assert_bit_equal(np.array(mg.getPosition()), want_pos, "getPosition()")
    assert mg.stats().host_syncs == steps
    mg.close()


@pytest.mark.gpu
@pytest.mark.parametrize("transport", ["loopback", "streams"])
def test_two_layer_hops_and_tiny_faces(transport):
    """z-velocities up to 2.5 cells per step: migrants land beyond the neighbour's first
    layer (the headers carry the near/far split); faces far too small for the traffic, so
    the exact-size second round carries most of it."""
    n, steps, world = 60000, 6, 3
    pos, vel = moving_state(n, 11, vz=25.0)
    settings = sph.default_settings(n, False)
    want, _ = single_domain(settings, pos, vel, steps)
    mg = M.MultiGpuSimulator(settings, world=world, transport=transport, face_capacity=300)
    mg.upload_state(pos, vel)
    for _ in range(steps):
        mg.simulate()
    got = mg.download_state()
    assert_bit_equal(got["pos"], want["pos"], "pos")
    assert_bit_equal(got["rho"], want["rho"], "rho")
    assert mg.stats().overflow_rounds == steps
    mg.close()


@pytest.mark.gpu
def test_hop_beyond_a_whole_slab_is_reported():
    """A particle that crosses more of the neighbour slab than the decomposition allows
    must end the run with an error, never a silently wrong result."""
    n, world = 40000, 8
    pos, vel = moving_state(n, 3, vz=1.0)
    vel[:50, 2] = 1500.0      # 15 cells per step: through a whole 12-layer slab
    pos[:50, 2] = 1.0
    settings = sph.default_settings(n, False)
    mg = M.MultiGpuSimulator(settings, world=world, transport="loopback")
    mg.upload_state(pos, vel)
    with pytest.raises(sph.SphError, match="crossed|neighbour|above|below"):
        for _ in range(4):
            mg.simulate()
        mg.sync()
    mg.close()


@pytest.mark.gpu
@pytest.mark.parametrize("sweep,transport", [("list", "loopback"), ("lds", "loopback"), ("list", "streams")])
def test_reference_initialiser_and_recut(sweep, transport):
    """-i random through 4 slabs with the cuts re-balanced every 3 steps (a re-cut is a
    stable filter of the rank-concatenated sequence, so it keeps the canonical order)."""
    n, steps, world = 262144, 7, 4
    settings = sph.default_settings(n, True)
    want, _ = single_domain(settings, None, None, steps, sweep=sweep)
    mg = M.MultiGpuSimulator(settings, world=world, transport=transport, sweep=sweep, recut_every=3)
    mg.setup()
    t = sph.Times()
    for _ in range(steps):
        mg.simulateAndTime(t)
    got = mg.download_state()
    assert_bit_equal(got["pos"], want["pos"], "pos")
    assert_bit_equal(got["vel"], want["vel"], "vel")
    assert t.iters == steps and t.sphUpdate > 0
    mg.close()


@pytest.mark.gpu
@pytest.mark.parametrize("transport", ["loopback", "streams"])
def test_get_position_right_after_a_recut_step(transport):
    """getPosition() after a step that ended with a re-cut (step % recut_every == 0): the re-cut
    re-uploads every slab through the pinned read-back buffer, so the rows must be fetched
    again (this used to return the previous frame, or zeros)."""
    n, world = 80000, 4
    pos, vel = moving_state(n, 7)
    settings = sph.default_settings(n, False)
    mg = M.MultiGpuSimulator(settings, world=world, transport=transport, recut_every=3)
    mg.upload_state(pos, vel)
    sim = sph.Simulator(settings)
    sim.upload_state(pos, vel)
    for step in range(1, 8):
        mg.simulate()
        sim.simulate()
        assert_bit_equal(np.array(mg.getPosition()), np.array(sim.getPosition()), f"getPosition() after step {step}")
    sim.close()
    mg.close()


@pytest.mark.gpu
def test_fresh_state_after_a_failed_step():
    """A step that fails (here: a particle hops across a whole slab) leaves the driver mid-step;
    upload_state()/setup() must make it usable again (the phase counter stayed stuck)."""
    n, world = 60000, 4
    pos, vel = moving_state(n, 3, vz=0.0)
    bad = vel.copy()
    bad[:, 2] = 400.0    # 40 cells per step: far beyond the one-layer halo
    settings = sph.default_settings(n, False)
    mg = M.MultiGpuSimulator(settings, world=world, transport="loopback")
    mg.upload_state(pos, bad)
    with pytest.raises(sph.SphError):
        for _ in range(3):
            mg.simulate()
        mg.sync()
    want, want_pos = single_domain(settings, pos, vel, 4)
    mg.upload_state(pos, vel)
    for _ in range(4):
        mg.simulate()
    assert_bit_equal(mg.download_state()["pos"], want["pos"], "pos after recovery")
    mg.close()


@pytest.mark.gpu
@pytest.mark.parametrize("world,transport,recut", [(4, "loopback", 0), (3, "streams", 0), (4, "loopback", 2)])
def test_click_impulse_in_multi_gpu_mode(world, transport, recut):
    """simulate() with `mouseClicked` set (simulator.cu:482-489) over z-slabs: every slab applies
    kernelMoveParticles to the layers it owns through that step's grid -- the same particles get the
    same impulse as in the single domain (the click used to be dropped silently).  With a
    re-cut at the end of the clicked step the impulse still lands before the state is redistributed."""
    n = 80000
    pos, vel = moving_state(n, 9)
    settings = sph.default_settings(n, False)
    sim = sph.Simulator(settings)
    sim.upload_state(pos, vel)
    mg = M.MultiGpuSimulator(settings, world=world, transport=transport, recut_every=recut)
    mg.upload_state(pos, vel)
    clicks = {2: (400, 300), 4: (250, 200), 5: (590, 440)}
    for step in range(1, 8):
        if step in clicks:
            for s in (sim, mg):
                s.mouseClicked, s.clickCoords = True, clicks[step]
        sim.simulate()
        mg.simulate()
        assert not mg.mouseClicked
    want, got = sim.download_state(), mg.download_state()
    plain, _ = single_domain(settings, pos, vel, 7)
    assert (want["vel"] != plain["vel"]).any(axis=1).sum() > 50, "the clicks moved something"
    assert_bit_equal(got["vel"], want["vel"], "vel")
    assert_bit_equal(got["pos"], want["pos"], "pos")
    sim.close()
    mg.close()


@pytest.mark.gpu
def test_one_rank_failing_stops_its_neighbours_with_an_error_not_a_hang():
    """One process per GPU (here: one driver object per rank, mailbox transport, stepped phase by
    phase): a particle that crosses a whole slab in one step makes ONE rank's check fail.  That
    rank finishes the step's message rounds, says farewell (an exchange A whose header carries
    status = 1) and returns the error; its neighbours read the status in their next step, do the
    same and return an error too -- nobody is left waiting for a message (the mailbox would report
    "the sending rank has not run this phase yet", RCCL would hang).  A fresh state then runs."""
    n, world = 60000, 3
    pos, vel = moving_state(n, 21, vz=3.0)
    k = int(np.argmin(pos[:, 2]))
    vel[k] = (0.0, 0.0, 650.0)          # 65 cells per step: from slab 0 across slab 1 into slab 2
    settings = sph.default_settings(n, False)
    ranks = [M.MultiGpuSimulator(settings, world=world, rank=r, devices=[0], transport="mailbox") for r in range(world)]
    for mg in ranks:
        mg.upload_state(pos, vel)
    died = {}
    for step in range(1, 8):
        for phase in (1, 2, 3, 4):
            for r, mg in enumerate(ranks):
                if r in died:
                    continue
                try:
                    mg.step_phase(phase)
                except sph.SphError as e:
                    died[r] = (step, phase, str(e))
        if len(died) == world:
            break
    assert len(died) == world, died
    assert all(ph == 4 for _, ph, _ in died.values()), died          # errors surface at the END of a step
    assert not any("mailbox" in msg for _, _, msg in died.values()), died
    first = min(died.values())[0]
    origin = [r for r, d in died.items() if d[0] == first]
    assert origin == [1] and "crossed" in died[1][2], died
    assert died[0][0] == died[2][0] == first + 1 and "neighbour slab reported a failure" in died[0][2], died
    with pytest.raises(sph.SphError):                                # ... and keep being reported
        ranks[1].step_phase(1)
    # a fresh state: the same objects run again, bit-equal to the single domain
    pos2, vel2 = moving_state(n, 22, vz=8.0)
    want, _ = single_domain(settings, pos2, vel2, 3)
    for mg in ranks:
        mg.upload_state(pos2, vel2)
    for _ in range(3):
        for phase in (1, 2, 3, 4):
            for mg in ranks:
                mg.step_phase(phase)
    got = np.full((n, 3), np.nan, np.float32)
    for mg in ranks:
        d = mg.download_state()
        m_ = ~np.isnan(d["pos"][:, 0])
        got[m_] = d["pos"][m_]
    assert_bit_equal(got, want["pos"], "after the failed run: pos")
    for mg in ranks:
        mg.close()


@pytest.mark.gpu
def test_sinking_fluid_recut_moves_the_cuts():
    """Mass that drifts along z makes the static cuts lopsided; the re-cut follows it."""
    n, world = 60000, 4
    rng = np.random.default_rng(2)
    pos = rng.uniform(1.0, 9.0, (n, 3)).astype(np.float32)
    vel = np.zeros((n, 3), np.float32)
    vel[:, 2] = 6.0           # everything drifts towards +z, 0.6 cells per step
    settings = sph.default_settings(n, False)
    want, _ = single_domain(settings, pos, vel, 12)
    mg = M.MultiGpuSimulator(settings, world=world, transport="loopback", recut_every=4)
    mg.upload_state(pos, vel)
    for _ in range(12):
        mg.simulate()
    got = mg.download_state()
    assert_bit_equal(got["pos"], want["pos"], "pos")
    st = mg.stats()
    assert st.recuts >= 1
    owned = list(st.owned[:world])
    assert max(owned) - min(owned) < 0.25 * n / world, owned
    mg.close()


@pytest.mark.gpu
@pytest.mark.parametrize("world,face", [(2, 0), (4, 0), (3, 250)])
def test_one_object_per_rank_code_path(world, face):
    """The one-process-per-GPU code path (every driver object sees ONE slab; what its
    neighbours hold it only knows from their message headers) on a one-GPU box: `world`
    objects in this process stand for the ranks, messages go through the mailbox
    transport, every object is stepped phase by phase.  Bit-equal to the single domain."""
    n, steps = 60000, 6
    pos, vel = moving_state(n, 17, vz=14.0)
    settings = sph.default_settings(n, False)
    want, _ = single_domain(settings, pos, vel, steps)
    ranks = [M.MultiGpuSimulator(settings, world=world, rank=r, devices=[0], transport="mailbox",
                                 face_capacity=face) for r in range(world)]
    for mg in ranks:
        mg.upload_state(pos, vel)
    for _ in range(steps):
        for phase in (1, 2, 3, 4):
            for mg in ranks:
                mg.step_phase(phase)
    pos_got = np.full((n, 3), np.nan, np.float32)
    rho_got = np.full(n, np.nan, np.float32)
    total = 0
    for mg in ranks:
        d = mg.download_state()
        own = ~np.isnan(d["rho"])
        assert own.sum() == d["written"]
        assert np.isnan(rho_got[own]).all(), "a particle is owned by two ranks"
        pos_got[own] = d["pos"][own]
        rho_got[own] = d["rho"][own]
        total += d["written"]
        assert mg.stats().host_syncs == steps
    assert total == n
    assert_bit_equal(pos_got, want["pos"], "pos")
    assert_bit_equal(rho_got, want["rho"], "rho")
    for mg in ranks:
        mg.close()


@pytest.mark.gpu
@pytest.mark.parametrize("n,world", [(1, 2), (70, 3), (500, 8), (4097, 4)])
def test_tiny_and_nearly_empty_slabs(n, world):
    """Slabs that own no particle at all, waves with a handful of valid lanes."""
    rng = np.random.default_rng(n)
    pos = rng.uniform(0.2, 9.8, (n, 3)).astype(np.float32)
    vel = rng.uniform(-3, 3, (n, 3)).astype(np.float32)
    settings = sph.default_settings(n, False)
    want, _ = single_domain(settings, pos, vel, 5)
    mg = M.MultiGpuSimulator(settings, world=world, transport="loopback")
    mg.upload_state(pos, vel)
    for _ in range(5):
        mg.simulate()
    got = mg.download_state()
    assert got["written"] == n
    assert_bit_equal(got["pos"], want["pos"], "pos")
    assert_bit_equal(got["rho"], want["rho"], "rho")
    mg.close()
