"""Multi-slab path: N z-slabs must reproduce the single-domain result BIT FOR
BIT (canonical order is preserved across migration and halo merge).
CPU: the slab driver (product code) over an oracle-backed compute backend, in
one process (loopback) and over torch.distributed gloo with world_size 2 and 3.
GPU: the same driver over libsph_hip.so, several slabs on one device."""
import os
import sys

import numpy as np
import pytest
import torch

import cudafluidsimulator_amd as sph
import slab_rehearsal as S
from helpers import assert_bit_equal
from oracle import oracle as O
from slab_backend_oracle import OracleSlabBackend

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def moving_state(n, seed):
    """Random fluid with z-velocities up to ~0.9 cell/step so particles migrate
    between slabs every step (and bounce off the z walls)."""
    rng = np.random.default_rng(seed)
    pos = rng.uniform(0.5, 9.5, (n, 3)).astype(np.float32)
    vel = rng.uniform(-1, 1, (n, 3)).astype(np.float32)
    vel[:, 2] = rng.uniform(-9, 9, n).astype(np.float32)
    return pos, vel


def reference_run(pos, vel, steps):
    ref = O.OracleSim(len(pos), False)
    ref.upload(pos, vel)
    ref.step(steps)
    return ref.download()


def collect(slabs_or_parts, n):
    pos = np.full((n, 3), np.nan, np.float32)
    vel = np.full((n, 3), np.nan, np.float32)
    rho = np.full(n, np.nan, np.float32)
    seen = 0
    for p4, v4 in slabs_or_parts:
        p4 = np.asarray(p4); v4 = np.asarray(v4)
        ids = np.ascontiguousarray(p4[:, 3]).view(np.uint32)
        pos[ids] = p4[:, :3]; vel[ids] = v4[:, :3]; rho[ids] = v4[:, 3]
        seen += len(ids)
    assert seen == n and not np.isnan(pos).any()
    return pos, vel, rho


def test_partition_layers():
    hist = np.zeros(100, int); hist[10:90] = 1000
    b = S.partition_layers(hist, 8)
    assert b[0][0] == 0 and b[-1][1] == 100 and all(b[i][1] == b[i + 1][0] for i in range(7))
    counts = [hist[a:c].sum() for a, c in b]
    assert max(counts) - min(counts) <= 1000
    with pytest.raises(ValueError):
        S.partition_layers(hist, 60)
    lopsided = np.zeros(100, int); lopsided[50] = 10
    b = S.partition_layers(lopsided, 4)
    assert all(c - a >= 2 for a, c in b)


@pytest.mark.parametrize("face_cap", [None, 5])
@pytest.mark.parametrize("world", [2, 4])
def test_loopback_slabs_equal_single_domain_cpu(world, face_cap):
    """face_cap=5: every face outgrows its fixed-size message, so the exact-size
    overflow round carries most of the halo (both window layouts are exercised)."""
    n, steps = 6000, 12
    pos, vel = moving_state(n, 21)
    settings = sph.default_settings(n, False)
    p4, v4 = S.pack_state(pos, vel)
    bounds, parts = S.split_initial(p4, v4, settings.h, 100, world)
    slabs = []
    for r, ((zlo, zhi), (pp, vv)) in enumerate(zip(bounds, parts)):
        sl = S.Slab(OracleSlabBackend(settings, n), r, world, zlo, zhi, 100, face_cap=face_cap)
        sl.load(torch.from_numpy(pp), torch.from_numpy(vv))
        slabs.append(sl)
    S.run_loopback(slabs, steps)
    got = collect([tuple(x.numpy() for x in sl.owned()) for sl in slabs], n)
    want = reference_run(pos, vel, steps)
    assert_bit_equal(got[0], want["pos"], "pos")
    assert_bit_equal(got[1], want["vel"], "vel")
    assert_bit_equal(got[2], want["rho"], "rho")
    assert (sum(sl.overflows for sl in slabs) > 0) == (face_cap is not None)
    # migration really happened
    assert sum(sl.n_own for sl in slabs) == n
    assert any(sl.n_own != len(parts[i][0]) for i, sl in enumerate(slabs))


def _gloo_worker(rank, world, port, n, steps, seed, outdir, face_cap=None):
    import torch.distributed as dist
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="2")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pos, vel = moving_state(n, seed)
    settings = sph.default_settings(n, False)
    p4, v4 = S.pack_state(pos, vel)
    bounds, parts = S.split_initial(p4, v4, settings.h, 100, world)
    sl = S.Slab(OracleSlabBackend(settings, n), rank, world, *bounds[rank], 100, face_cap=face_cap)
    sl.load(torch.from_numpy(parts[rank][0]), torch.from_numpy(parts[rank][1]))
    tr = S.DistTransport(dist, rank, world, torch.device("cpu"))
    for _ in range(steps):
        S.step_distributed(sl, tr)
    p, v = sl.owned()
    np.savez(os.path.join(outdir, f"rank{rank}.npz"), pos4=p.numpy(), vel4=v.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,face_cap", [(2, None), (3, None), (3, 20)])
def test_gloo_slabs_equal_single_domain(world, face_cap, tmp_path):
    import torch.multiprocessing as mp
    n, steps, seed = 4000, 10, 33
    port = 29600 + world + (os.getpid() % 200) + (7 if face_cap else 0)
    mp.spawn(_gloo_worker, args=(world, port, n, steps, seed, str(tmp_path), face_cap), nprocs=world,
             join=True)
    parts = []
    for r in range(world):
        d = np.load(tmp_path / f"rank{r}.npz")
        parts.append((d["pos4"], d["vel4"]))
    got = collect(parts, n)
    pos, vel = moving_state(n, seed)
    want = reference_run(pos, vel, steps)
    assert_bit_equal(got[0], want["pos"], "pos")
    assert_bit_equal(got[1], want["vel"], "vel")


def _gloo_failing_worker(rank, world, port, outdir):
    import torch.distributed as dist
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="2")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n = 3000
    pos, vel = moving_state(n, 7)
    settings = sph.default_settings(n, False)
    p4, v4 = S.pack_state(pos, vel)
    bounds, parts = S.split_initial(p4, v4, settings.h, 100, world)
    sl = S.Slab(OracleSlabBackend(settings, n), rank, world, *bounds[rank], 100)
    sl.load(torch.from_numpy(parts[rank][0]), torch.from_numpy(parts[rank][1]))
    tr = S.DistTransport(dist, rank, world, torch.device("cpu"))
    if rank == 0:   # this rank's third step finds its capacity exceeded
        real = sl.assemble
        calls = [0]

        def failing():
            calls[0] += 1
            if calls[0] == 3:
                raise S.SphError("slab capacity exceeded by halo + migrants")
            real()
        sl.assemble = failing
    try:
        for _ in range(6):
            S.step_distributed(sl, tr)
        outcome = "finished"
    except S.SphError as e:
        outcome = "SphError: " + str(e)
    except Exception as e:     # the neighbour: its receive fails, it does not hang
        outcome = "transport: " + type(e).__name__
    with open(os.path.join(outdir, f"rank{rank}.txt"), "w") as f:
        f.write(outcome)


@pytest.mark.timeout(180)
def test_gloo_failing_rank_does_not_leave_its_neighbour_waiting(tmp_path):
    """a rank whose capacity / halo check fails must not leave its peers blocked
    in batch_isend_irecv: it closes its end of the transport before raising."""
    import torch.multiprocessing as mp
    port = 29900 + (os.getpid() % 90)
    mp.spawn(_gloo_failing_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0 = (tmp_path / "rank0.txt").read_text()
    r1 = (tmp_path / "rank1.txt").read_text()
    assert r0.startswith("SphError: slab capacity exceeded")
    assert r1.startswith("transport: "), r1


@pytest.mark.gpu
@pytest.mark.parametrize("world,sweep,face_cap", [(2, "list", None), (3, "list", None), (2, "lds", None),
                                                  (3, "lds", None), (3, "list", 300)])
def test_hip_loopback_slabs_equal_single_domain(world, sweep, face_cap):
    n, steps = 60000, 8
    pos, vel = moving_state(n, 5)
    settings = sph.default_settings(n, False)
    p4, v4 = S.pack_state(pos, vel)
    bounds, parts = S.split_initial(p4, v4, settings.h, 100, world)
    slabs = []
    for r, ((zlo, zhi), (pp, vv)) in enumerate(zip(bounds, parts)):
        sl = S.Slab(S.HipSlabBackend(settings, n, device=0, sweep=sweep), r, world, zlo, zhi, 100,
                    face_cap=face_cap)
        sl.load(torch.from_numpy(pp).cuda(), torch.from_numpy(vv).cuda())
        slabs.append(sl)
    S.run_loopback(slabs, steps)
    got = collect([tuple(x.cpu().numpy() for x in sl.owned()) for sl in slabs], n)
    want = reference_run(pos, vel, steps)
    assert_bit_equal(got[0], want["pos"], "pos")
    assert_bit_equal(got[1], want["vel"], "vel")
    assert_bit_equal(got[2], want["rho"], "rho")
    assert (sum(sl.overflows for sl in slabs) > 0) == (face_cap is not None)
    # and against the single-domain HIP path
    sim = sph.Simulator(settings)
    sim.upload_state(pos, vel)
    for _ in range(steps):
        sim.simulate()
    assert_bit_equal(got[0], sim.download_state()["pos"], "vs single-domain HIP")
    sim.close()
    for sl in slabs:
        sl.b.close()


@pytest.mark.gpu
def test_hip_slabs_exchanging_through_rccl_self_send():
    """The slab messages -- the very tensor views DistTransport posts -- through
    torch.distributed's RCCL point-to-point API: a world of ONE rank sending to itself
    (all a one-GPU box allows; >1 rank per GPU is refused by RCCL)."""
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", str(29650 + os.getpid() % 200))
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        n, steps, world = 60000, 6, 3
        pos, vel = moving_state(n, 9)
        settings = sph.default_settings(n, False)
        p4, v4 = S.pack_state(pos, vel)
        bounds, parts = S.split_initial(p4, v4, settings.h, 100, world)
        slabs = []
        for r, ((zlo, zhi), (pp, vv)) in enumerate(zip(bounds, parts)):
            sl = S.Slab(S.HipSlabBackend(settings, n, device=0), r, world, zlo, zhi, 100)
            sl.load(torch.from_numpy(pp).cuda(), torch.from_numpy(vv).cuda())
            slabs.append(sl)
        S.run_loopback(slabs, steps, copy=S.nccl_self_copy(dist))
        torch.cuda.synchronize()
        got = collect([tuple(x.cpu().numpy() for x in sl.owned()) for sl in slabs], n)
        want = reference_run(pos, vel, steps)
        assert_bit_equal(got[0], want["pos"], "pos")
        assert_bit_equal(got[1], want["vel"], "vel")
        for sl in slabs:
            sl.b.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
def test_hip_slab_reference_initialiser_4_slabs():
    n, steps, world = 262144, 6, 4
    settings = sph.default_settings(n, True)
    p4, v4 = S.make_initial(settings)
    bounds, parts = S.split_initial(p4, v4, settings.h, 100, world)
    slabs = []
    for r, ((zlo, zhi), (pp, vv)) in enumerate(zip(bounds, parts)):
        sl = S.Slab(S.HipSlabBackend(settings, n, device=0), r, world, zlo, zhi, 100)
        sl.load(torch.from_numpy(pp).cuda(), torch.from_numpy(vv).cuda())
        slabs.append(sl)
    S.run_loopback(slabs, steps)
    got = collect([tuple(x.cpu().numpy() for x in sl.owned()) for sl in slabs], n)
    sim = sph.Simulator(settings)
    sim.setup()
    for _ in range(steps):
        sim.simulate()
    assert_bit_equal(got[0], sim.download_state()["pos"], "slabs vs single domain")
    sim.close()
    for sl in slabs:
        sl.b.close()


@pytest.mark.gpu
def test_hip_slab_config4_size_16M_particles_4_slabs():
    """BASELINE config 4 at FULL size (-n 16777216 -i random, 4 z-slabs): the
    decomposition (capacity per slab, not n; one-layer halos of ~210 K particles;
    fixed-size exchange messages) against the single-domain HIP path, which the
    headline checksums pin to the oracle.  Two steps, bit for bit."""
    n, steps, world = 16777216, 2, 4
    settings = sph.default_settings(n, True)
    p4, v4 = S.make_initial(settings)
    bounds, parts = S.split_initial(p4, v4, settings.h, 100, world)
    cap = int(max(len(p[0]) for p in parts) * 1.3) + 65536
    face = min(S.default_face_cap(p4, settings.h, 100), cap)
    slabs = []
    for r, ((zlo, zhi), (pp, vv)) in enumerate(zip(bounds, parts)):
        sl = S.Slab(S.HipSlabBackend(settings, cap, device=0), r, world, zlo, zhi, 100, face_cap=face)
        sl.load(torch.from_numpy(pp).cuda(), torch.from_numpy(vv).cuda())
        slabs.append(sl)
    del parts
    S.run_loopback(slabs, steps)
    got = collect([tuple(x.cpu().numpy() for x in sl.owned()) for sl in slabs], n)
    assert sum(sl.overflows for sl in slabs) == 0
    for sl in slabs:
        sl.b.close()
    sim = sph.Simulator(settings)
    sim.setup()
    for _ in range(steps):
        sim.simulate()
    st = sim.download_state()
    sim.close()
    assert_bit_equal(got[0], st["pos"], "16.7M particles, 4 slabs vs single domain: pos")
    assert_bit_equal(got[2], st["rho"], "16.7M particles, 4 slabs vs single domain: rho")


@pytest.mark.gpu
def test_out_of_grid_positions_are_reported_like_the_reference(capfd):
    """getGridCell printf's "OOB particle: x = ..." and the position for a cell outside the table
    (simulator.cu:60-73).  Positions that came through setup()/upload_state() and the integrator's
    wall clamp never are; caller-owned slab buffers can hold anything: the first sort pass logs such
    rows, clamps their cell into the table (no out-of-bounds index), and the host prints the
    reference's two lines at its next synchronisation."""
    n = 5000
    settings = sph.default_settings(n, False)
    be = S.HipSlabBackend(settings, n + 64)
    rng = np.random.default_rng(1)
    pos = np.zeros((n, 4), np.float32)
    pos[:, :3] = rng.uniform(1.0, 9.0, (n, 3))
    pos[:, 3] = np.arange(n, dtype=np.uint32).view(np.float32)
    pos[17, 0] = -0.15      # cell -1
    pos[4000, 2] = 10.55    # cell 105 of 100
    be.pos[0][:n] = torch.from_numpy(pos).to(be.device)
    be.sort(0, 0, n, [])
    torch.cuda.synchronize()
    be._check(be._L.sph_sync(be._h), "sph_sync")
    out = capfd.readouterr().out
    assert "OOB particle: x = -1" in out and "OOB particle: z = 105" in out, out
    assert "(-0.150000," in out and "10.550000)" in out
    # clamped, not dropped: the sorted stream still holds every row
    ids = be.pos[1][:n, 3].cpu().numpy().view(np.uint32)
    assert np.array_equal(np.sort(ids), np.arange(n, dtype=np.uint32))
    be.close()
