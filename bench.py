#!/usr/bin/env python3
"""Benchmark of the SPH step path: particle-steps/s of the `-m time` loop
(main.cpp:68-76: N x simulateAndTime from the reference initial condition) plus
the computeDensity roofline figures and the CPU-oracle baseline.

    python bench.py --gpus N --steps K --warmup W
    (N>1: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...)

A "step" is one simulateAndTime(): grid build + density + force/integrate +
position read-back.  Warm-up steps run first, then the initial condition is
re-uploaded so the K timed steps are steps 1..K of the reference's run (work per
step grows as the fluid settles).  Prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
VALU_PEAK_LANEOPS = 7.86e13    # 256 CU x 4 SIMD x 32 lanes x 2.4 GHz
DENSITY_BYTES_PER_PARTICLE = 20  # read 12-B position, write rho + p
DENSITY_LANEOPS_PER_TEST = 12    # SURVEY.md 8d convention (per candidate pair test)
FORCE_LANEOPS_PER_PAIR = 55      # SURVEY.md 8d convention (per evaluated pair body)
FULL_RUN_STEPS = 100             # main.cpp:69: the -m time loop is 100 x simulateAndTime


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("-n", "--particles", type=int, default=4194304,
                    help="BASELINE.json configs[2]: -n 4194304 -i random -m time")
    ap.add_argument("--init", choices=["random", "grid"], default="random")
    ap.add_argument("--sweep", choices=["list", "lds", "direct", "linked"], default="list")
    ap.add_argument("--key", choices=["flattened", "morton"], default="flattened",
                    help="cell key of the grid build's sort; morton: --sweep direct only (ordering A/B)")
    ap.add_argument("--math", choices=["strict", "fast"], default="strict",
                    help="strict: bit-identical to the oracle (default); fast: FMA + approximate "
                         "rcp/rsq, tolerance-checked")
    ap.add_argument("--mode", choices=["time", "free", "display"], default="time",
                    help="time: simulateAndTime loop (-m time); free: simulate() loop; display: "
                         "simulate() + getPosition() every frame like display.cpp:36-37")
    ap.add_argument("--readback", choices=["copy", "mapped", "none"], default="copy",
                    help="copy: overlapped device->host copy per step (default); mapped: the force sweep "
                         "writes getPosition()'s buffer in host-mapped memory (zero-copy); none: kernel studies "
                         "only (SPH_FLAG_NO_READBACK: not the reference's step, never `value`)")
    ap.add_argument("--no-linked-leg", action="store_true",
                    help="skip the secondary run of the reference's linked-list neighbour structure")
    ap.add_argument("--no-fast-leg", action="store_true",
                    help="skip the extra SPH_MATH_FAST measurement")
    ap.add_argument("--loopback-slabs", type=int, default=0,
                    help="N > 1: step the domain as N z-slabs on ONE GPU through the C++ multi-GPU driver's "
                         "loopback transport (what the decomposition costs per slab; never `value` of a "
                         "multi-GPU run)")
    ap.add_argument("--no-extra-legs", action="store_true",
                    help="skip the full 100-step run (when --steps is not 100) and the config 2 / config 4 legs")
    ap.add_argument("--no-count-replay", action="store_true",
                    help="skip the untimed replay that counts pair tests / hits (valu_frac figures)")
    ap.add_argument("--cpu-steps", type=int, default=12,
                    help="oracle steps timed for cpu_baseline (0 = skip)")
    ap.add_argument("--cpu-particles", type=int, default=0, help="0 = same n as the GPU run")
    return ap.parse_args()


def cpu_baseline(n, random_init, steps):
    """The CPU oracle (kind 'port': the reference has no CPU path and its CUDA
    source cannot be built here) on the host's cores, bounded sample.  A GPU box
    gives a job a CPU share smaller than the core count it reports, so two team
    sizes are timed -- 16 threads and every reported core -- and the faster one is
    the baseline (`cores` = the threads of that run)."""
    from oracle import oracle as O
    build = O.use_native_build()  # -O3 -march=native -fopenmp, built on this host

    def run(threads):
        O.set_num_threads(threads)
        sim = O.OracleSim(n, random_init)
        sim.setup()
        t0 = time.perf_counter()
        sim.step(steps)
        dt = time.perf_counter() - t0
        sim.close()
        return dt

    ncpu = os.cpu_count() or 1
    legs = {t: run(t) for t in sorted({min(16, ncpu), ncpu})}
    threads, dt = min(legs.items(), key=lambda kv: kv[1])
    out = {"value": n * steps / dt, "unit": "particle-steps/s", "cores": threads,
           "kind": "port", "steps": steps, "build": build,
           "sample": f"first {steps} steps of -n {n} -i {'random' if random_init else 'grid'} "
                     f"(of the 100-step run; later steps cost up to 4.7x more), "
                     f"OpenMP oracle ({build}), {dt:.1f} s; team sizes tried: "
                     + ", ".join(f"{t} threads {n * steps / d:.3g}/s" for t, d in legs.items())}
    # the same oracle on ONE core (bounded: the first step only)
    O.set_num_threads(1)
    sim = O.OracleSim(n, random_init)
    sim.setup()
    t0 = time.perf_counter()
    sim.step(1)
    dt1 = time.perf_counter() - t0
    sim.close()
    O.set_num_threads(threads)
    out["single_thread"] = {"value": n / dt1, "unit": "particle-steps/s", "cores": 1,
                            "sample": f"first step only, {dt1:.1f} s"}
    return out


def timed_run(sph, _lib, torch, s, args, K, W, device, settle=0, sweep=None, math=None, mode=None):
    """K x simulateAndTime (or simulate) from the reference initial condition, after `settle` +
    W untimed steps and a re-upload of the initial condition.  Returns elapsed seconds, the
    accumulated HIP-event kernel times, the -m time table and the wall time after every step."""
    sweep = sweep or args.sweep
    math = math or args.math
    mode = mode or args.mode
    rb_flag = {"mapped": _lib.SPH_FLAG_MAPPED_POSITIONS, "none": _lib.SPH_FLAG_NO_READBACK}.get(args.readback, 0)

    def one_step(sm, tm):
        if mode == "time":
            sm.simulateAndTime(tm)
        else:
            sm.simulate()
            if mode == "display":
                sm.getPosition()  # blocks until this frame's positions are on the host
    sim = sph.Simulator(s, sweep=sweep, flags=rb_flag, device=device, math=math, key_order=args.key)
    sim.setup()
    times = sph.Times()
    # Runtime settle, NOT part of W: with torch's bundled HIP runtime loaded, a process's 4th-7th
    # step meets a one-off ~7 ms stall of the GPU's queues (scripts/studies/early_stall.py; never
    # with the system runtime ./sph links) -- a fifth of a 20-step run at small n.  Twelve
    # untimed steps take it out of the way whatever W is; then the W warm-up steps proper.
    for _ in range(settle):
        one_step(sim, times)
    sim.sync()
    if settle:
        sim.setup()
    for _ in range(W):
        one_step(sim, times)
    sim.sync()
    sim.setup()  # back to the initial condition: timed steps are steps 1..K
    sim.kernel_times(reset=True)
    times = sph.Times()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    walls = []
    for _ in range(K):
        one_step(sim, times)
        walls.append(time.perf_counter() - t0)
    sim.sync()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if os.environ.get("SPH_BENCH_STEP_WALLS"):   # diagnostic: per-step wall times on stderr
        print("step walls (ms): " + " ".join("%.2f" % ((w - p0) * 1e3) for w, p0 in zip(walls, [0.0] + walls[:-1])),
              file=sys.stderr)
    out = dict(elapsed=elapsed, kt=sim.kernel_times(), times=times, walls=walls)
    if os.environ.get("SPH_STAMPS"):
        out["stamps"] = sim.debug_counters()
    sim.close()
    return out


def count_replay(sph, _lib, s, args, K, device):
    """Untimed replay of the same K steps with the counters on (the run is deterministic: same
    trajectory, same pair tests and hits per step)."""
    csim = sph.Simulator(s, sweep=args.sweep, flags=_lib.SPH_FLAG_COUNT_PAIRS, device=device,
                         math=args.math, key_order=args.key)
    csim.setup()
    for _ in range(K):
        csim.simulate()
    ckt = csim.kernel_times()
    st = max(int(ckt.steps), 1)
    out = {"pair_tests": ckt.pair_tests / st, "pair_hits": ckt.pair_hits / st,
           "pair_hits_recorded": csim.debug_counters()[15] / st}
    csim.close()
    return out


def roofline_of(n_local, kt, sweep, counts):
    """SURVEY.md 8d figures of one run: computeDensity against the HBM roof (algorithmic bytes /
    average launch time from HIP events on the launch stream) and, because the sweeps are not
    HBM-bound, the VALU fractions next to it."""
    steps = max(int(kt.steps), 1)
    dens_s, force_s = kt.density / steps, kt.force / steps
    achieved = DENSITY_BYTES_PER_PARTICLE * n_local / dens_s / 1e9 if dens_s > 0 else 0.0
    roof = {"bound": "hbm",
            "kernel": {"lds": "k_density_lds", "direct": "k_density_direct", "linked": "k_density_linked",
                       "list": "k_density_mask_lds"}[sweep] + " (computeDensity)",
            "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
            "traffic": None, "avg_launch_us": dens_s * 1e6,
            "algorithmic_bytes_per_launch": DENSITY_BYTES_PER_PARTICLE * n_local,
            "note": "the sweep is VALU-bound, not HBM-bound (SURVEY.md 8d): see valu_frac"}
    pairs = (counts or {}).get("pair_tests") or None
    if pairs:
        roof["pair_tests_per_launch"] = pairs
        roof["valu_frac"] = pairs * DENSITY_LANEOPS_PER_TEST / dens_s / VALU_PEAK_LANEOPS
        roof["valu_frac_note"] = (f"pair tests (counted on the GPU) x {DENSITY_LANEOPS_PER_TEST} lane-ops "
                                  f"(SURVEY.md 8d convention) / launch time / {VALU_PEAK_LANEOPS:.3g} lane-ops/s "
                                  "(256 CU x 4 SIMD x 32 lanes x 2.4 GHz)")
        bodies = counts.get("pair_hits") or 0
        recorded = counts.get("pair_hits_recorded") or 0
        if sweep == "list" and recorded:
            # the list sweep evaluates the pair body for recorded hits only, minus the pairs its
            # zero-pair filter drops (no pressure on either side, same velocity: exactly +-0)
            roof["force_hits_recorded_per_launch"] = recorded
            roof["force_pair_bodies_per_launch"] = bodies
            roof["force_hit_fraction"] = recorded / pairs
            roof["force_zero_pairs_dropped"] = 1.0 - bodies / recorded
            roof["force_valu_frac"] = bodies * FORCE_LANEOPS_PER_PAIR / force_s / VALU_PEAK_LANEOPS
            roof["force_valu_frac_note"] = (
                f"k_force_dealt: pair bodies actually evaluated (popcount of the hit masks after the zero-pair "
                f"filter, counted on the GPU) x {FORCE_LANEOPS_PER_PAIR} lane-ops (SURVEY.md 8d convention) / "
                f"launch time / VALU peak")
        elif sweep in ("lds", "direct"):
            # these sweeps run the test for every candidate and the body under a mask
            roof["force_valu_frac"] = pairs * FORCE_LANEOPS_PER_PAIR / force_s / VALU_PEAK_LANEOPS
            roof["force_valu_frac_note"] = "every candidate priced at the body's 55 lane-ops (tested, body masked)"
    return roof


def kernel_ms(kt):
    steps = max(int(kt.steps), 1)
    return {"hash": kt.hash / steps * 1e3, "sort": kt.sort / steps * 1e3, "gather_cells": kt.gather / steps * 1e3,
            "density": kt.density / steps * 1e3, "force_integrate": kt.force / steps * 1e3,
            "readback_d2h": kt.readback / steps * 1e3}


RUNTIME_SETTLE_STEPS = 12   # see main(): untimed, before the W warm-up steps


def run_mgpu_bench(args, dist, rank, world, local_rank):
    """N > 1: the C++ in-process driver (libsph_mgpu.so), one process per GPU: each rank
    drives ONE z-slab; the halo layers travel by ncclSend/ncclRecv between the ranks' own
    RCCL communicator (unique id made on rank 0, handed out through torch.distributed).
    --loopback-slabs N: one process, N slabs on one GPU, device-to-device copies."""
    import torch
    import cudafluidsimulator_amd as sph
    from cudafluidsimulator_amd import mgpu as M
    n = args.particles
    settings = sph.default_settings(n, args.init == "random")
    if dist is not None:
        uid = torch.zeros(128, dtype=torch.uint8, device="cuda")
        if rank == 0:
            uid = torch.tensor(list(M.unique_id()), dtype=torch.uint8, device="cuda")
        dist.broadcast(uid, src=0)
        mg = M.MultiGpuSimulator(settings, world=world, rank=rank, devices=[local_rank],
                                 unique_id=bytes(uid.cpu().tolist()), transport="rccl", sweep=args.sweep,
                                 math=args.math)
    else:
        mg = M.MultiGpuSimulator(settings, world=args.loopback_slabs, transport="loopback",
                                 devices=[local_rank], sweep=args.sweep, math=args.math)
    mg.setup()
    times = sph.Times()
    # at least one untimed step: the first exchange builds the RCCL channels
    for _ in range(RUNTIME_SETTLE_STEPS):
        mg.simulateAndTime(times) if args.mode == "time" else mg.simulate()
    mg.sync()
    mg.setup()
    for _ in range(max(args.warmup, 1)):
        mg.simulateAndTime(times) if args.mode == "time" else mg.simulate()
    mg.sync()
    mg.setup()
    mg.stats(reset=True)
    times = sph.Times()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        mg.simulateAndTime(times) if args.mode == "time" else mg.simulate()
    mg.sync()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    st = mg.stats()
    local = st.local_slabs
    from cudafluidsimulator_amd import SphKernelTimes
    kt = SphKernelTimes()   # slab 0 of this process stands for the rank in the roofline figures
    kt.sort, kt.density, kt.force, kt.steps = st.grid_s[0], st.density_s[0], st.force_s[0], st.steps
    out = dict(elapsed=elapsed, n_total=n, n_local=int(st.owned[0]), times=times, kt=kt,
               mgpu=dict(slabs=args.loopback_slabs if dist is None else world,
                         host_syncs_per_step=st.host_syncs / max(st.steps, 1),
                         overflow_rounds=int(st.overflow_rounds),
                         kernel_ms_per_step_per_slab=[st.kernel_s[k] / max(st.steps, 1) * 1e3 for k in range(local)],
                         owned=[int(st.owned[k]) for k in range(local)]))
    mg.close()
    return out


def load_traffic(n, args, steps):
    """HBM bytes per computeDensity launch from the committed rocprofv3 PMC passes
    (FETCH_SIZE and WRITE_SIZE collected in separate --pmc runs of this very
    command line; FETCH_SIZE doubled per MI355X_MICROARCH.md's gfx950 note for
    16-B/lane streaming reads).  Work per launch grows over the run, so only a
    measurement of the SAME step count K (and warm-up) is used: None otherwise --
    windows are never mixed."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    if not os.path.exists(path):
        return None
    try:
        for e in json.load(open(path)):
            if (e["n"] == n and e["init"] == args.init and e["sweep"] == args.sweep and e["gpus"] == 1
                    and e.get("math", "strict") == args.math and e.get("steps") == steps
                    and e.get("current", True)):
                return e
    except Exception:
        return None
    return None


def main():
    args = parse()
    # stdout carries exactly ONE line (the JSON): whatever libraries print on file
    # descriptor 1 (the RCCL start-up banner, gloo connection notes) goes to stderr.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    force_mgpu = bool(os.environ.get("SPH_BENCH_FORCE_MGPU"))  # tests: the N>1 code path with one rank
    if world > 1 or force_mgpu:
        # The >= 2-rank RCCL path has never run on hardware: a rank that waits for ever
        # in a receive must not hold the node -- give up loudly after ten minutes instead.
        import signal

        def _give_up(signum, frame):
            sys.stderr.write("bench.py: multi-GPU run exceeded 600 s: giving up (a rank is waiting in an exchange?)\n")
            sys.stderr.flush()
            os._exit(3)
        signal.signal(signal.SIGALRM, _give_up)
        signal.alarm(600)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py: --gpus N>1 must be launched through torch.distributed.run")
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU (there is no CPU fallback); use -m 'not gpu' tests on CPU")
    # torch's lazy device initialisation (context, streams, allocator) happens HERE, not in the
    # torch.cuda.synchronize() that brackets the timed loop: its deferred work stalled the GPU's
    # queues for ~7 ms a few hundred microseconds into the loop (scripts/studies)
    torch.cuda.init()
    torch.cuda.synchronize()

    import cudafluidsimulator_amd as sph
    from cudafluidsimulator_amd import _lib

    n = args.particles
    random_init = args.init == "random"
    K, W = args.steps, args.warmup

    if world > 1 or args.loopback_slabs > 1 or force_mgpu:
        result = run_mgpu_bench(args, dist if (world > 1 or force_mgpu) else None, rank, world, local_rank)
    else:
        s = sph.default_settings(n, random_init)
        # the timed run carries no counting code at all (SPH_FLAG_COUNT_PAIRS adds atomics
        # to the density sweep); pair tests and hits come from an untimed replay below
        result = timed_run(sph, _lib, torch, s, args, K, W, local_rank, settle=RUNTIME_SETTLE_STEPS)
        result["n_total"] = n
        if not args.no_count_replay:
            result.update(count_replay(sph, _lib, s, args, K, local_rank))
        if args.sweep == "list" and args.math == "strict" and not args.no_linked_leg:
            # secondary figure: the reference's own neighbour structure (per-cell
            # linked lists, no sort) on this GPU, same K steps (not `value`)
            lsim = sph.Simulator(s, sweep="linked", device=local_rank)
            lsim.setup()
            lt = sph.Times()
            lsim.simulateAndTime(lt)
            lsim.sync()
            lsim.setup()
            lsim.kernel_times(reset=True)
            lt = sph.Times()
            torch.cuda.synchronize()
            l0 = time.perf_counter()
            for _ in range(K):
                lsim.simulateAndTime(lt) if args.mode == "time" else lsim.simulate()
            lsim.sync()
            result["linked_elapsed"] = time.perf_counter() - l0
            result["linked_kt"] = lsim.kernel_times()
            lsim.close()
        if args.math == "strict" and args.sweep not in ("direct", "linked") and not args.no_fast_leg:
            # secondary figure: the same K steps in SPH_MATH_FAST (not `value`)
            fsim = sph.Simulator(s, sweep=args.sweep, device=local_rank, math="fast")
            fsim.setup()
            ft = sph.Times()
            for _ in range(min(W, 3)):
                fsim.simulateAndTime(ft)
            fsim.sync()
            fsim.setup()
            torch.cuda.synchronize()
            f0 = time.perf_counter()
            for _ in range(K):
                fsim.simulateAndTime(ft) if args.mode == "time" else fsim.simulate()
            fsim.sync()
            result["fast_elapsed"] = time.perf_counter() - f0
            fsim.close()
        if not args.no_extra_legs and args.mode == "time" and args.readback == "copy":
            # The metric BASELINE.json names is the 100-step loop of main.cpp:68-76.  Whatever K the
            # command line asked for, the full loop runs too (0.2 s) and is reported as `full_run_100`.
            if K == FULL_RUN_STEPS:
                result["full100"] = dict(result)
            else:
                r100 = timed_run(sph, _lib, torch, s, args, FULL_RUN_STEPS, min(W, 5), local_rank)
                if not args.no_count_replay:
                    r100.update(count_replay(sph, _lib, s, args, FULL_RUN_STEPS, local_rank))
                r100["n_total"] = n
                result["full100"] = r100
            # one-line legs for the other single-GPU configurations of BASELINE.json (defaults only)
            if n == 4194304 and random_init and args.sweep == "list" and args.math == "strict":
                legs = {}
                for name, (ln, lk, lw, lrand, lsettle) in {
                        "config2_n262144_100steps": (262144, 100, 5, True, RUNTIME_SETTLE_STEPS),
                        "config4_n16777216_single_gpu_10steps": (16777216, 10, 2, True, RUNTIME_SETTLE_STEPS),
                        "config5_n67108864_grid_single_gpu_3steps": (67108864, 3, 1, False, 0)}.items():
                    ls = sph.default_settings(ln, lrand)
                    lr = timed_run(sph, _lib, torch, ls, args, lk, lw, local_rank, settle=lsettle)
                    init_name = "random" if lrand else "grid (dense-lattice extension beyond 109^3, DESIGN.md section 8)"
                    legs[name] = {"workload": f"-n {ln} -i {init_name} -m time, first {lk} steps" if lk != 100 else
                                  f"-n {ln} -i {init_name} -m time (100 steps)",
                                  "value": ln * lk / lr["elapsed"], "unit": "particle-steps/s",
                                  "ms_per_step": lr["elapsed"] / lk * 1e3, "steps": lk, "warmup": lw,
                                  "kernel_ms_per_step": kernel_ms(lr["kt"])}
                result["legs"] = legs

    if world > 1 or force_mgpu:
        t = torch.tensor([result["elapsed"]], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        result["elapsed"] = float(t.item())

    if rank == 0:
        elapsed = result["elapsed"]
        kt = result.get("kt")
        if kt is None:  # the C++ multi-GPU driver reports per-slab kernel time only
            from cudafluidsimulator_amd import SphKernelTimes
            kt = SphKernelTimes()
        n_local = result.get("n_local", n)
        roof = roofline_of(n_local, kt, args.sweep, result)
        full = K == FULL_RUN_STEPS
        out = {
            "metric": ("particle-steps/sec (100-step -m time)" if full else
                       f"particle-steps/sec (first {K} of the 100 steps of -m time)"),
            "full_run": full,
            "full_run_note": None if full else (
                f"steps 1..{K} only: work per step grows 4.7x over the 100 steps (pressure and floor "
                "contact start at step ~45), so this is not the 100-step figure: that one is `full_run_100` "
                "below, measured in this same process"),
            "value": result["n_total"] * K / elapsed,
            "unit": "particle-steps/s",
            "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": elapsed / K * 1e3,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"-n {result['n_total']} -i {args.init} -m {args.mode}, "
                                   f"{world}xMI355X, {args.key}-index radix sort + float4 SoA, sweep={args.sweep}, "
                                   + ("strict fp32 (bit-identical to the CPU oracle)" if args.math == "strict"
                                      else "FAST fp32 math (FMA, approximate rcp/rsq; 1e-5 tolerance mode)"),
                       "sweep": args.sweep,
                       "parallelism": "single domain" if world == 1 else
                       f"z-slabs x{world}, one process per GPU, C++ driver + RCCL send/recv halo"},
            "roofline": roof,
            "kernel_ms_per_step": kernel_ms(kt),
        }
        if "full100" in result:
            f = result["full100"]
            froof = roofline_of(n, f["kt"], args.sweep, f)
            ftr = load_traffic(n, args, steps=FULL_RUN_STEPS) if world == 1 else None
            if ftr:
                froof["traffic"] = ftr["bytes_per_launch"]
                froof["traffic_source"] = ftr["source"]
            out["full_run_100"] = {
                "metric": "particle-steps/sec (100-step -m time)", "value": n * FULL_RUN_STEPS / f["elapsed"],
                "unit": "particle-steps/s", "ms_per_step": f["elapsed"] / FULL_RUN_STEPS * 1e3,
                "steps": FULL_RUN_STEPS, "kernel_ms_per_step": kernel_ms(f["kt"]), "roofline": froof,
                "note": "the full loop of main.cpp:68-76 from the reference initial condition, run in this same "
                        "process whatever --steps is (`value` above is the --steps window)"}
        if "legs" in result:
            out["other_configs"] = result["legs"]
        if "times" in result:
            t = result["times"]
            out["m_time_table_s"] = {"grid_construction": t.buildGrid, "sph_update": t.sphUpdate,
                                     "data_transfer_exposed": t.memcpy}
        if "fast_elapsed" in result:
            out["fast_math"] = {"value": result["n_total"] * K / result["fast_elapsed"],
                                "unit": "particle-steps/s",
                                "note": "SPH_MATH_FAST (FMA + approximate rcp/rsq): not bit-exact; max relative "
                                        "position error 8.1e-7 after 100 steps of -n 8192 -i grid vs the oracle "
                                        "(north-star tolerance 1e-5; tests/test_gpu_parity.py). Not `value`."}
        if "linked_elapsed" in result:
            lk = result["linked_kt"]
            lst = max(int(lk.steps), 1)
            out["reference_structure"] = {
                "value": result["n_total"] * K / result["linked_elapsed"], "unit": "particle-steps/s",
                "ms_per_step": result["linked_elapsed"] / K * 1e3,
                "kernel_ms_per_step": {"grid": (lk.hash + lk.sort + lk.gather) / lst * 1e3,
                                       "density": lk.density / lst * 1e3,
                                       "force_integrate": lk.force / lst * 1e3},
                "note": "SPH_SWEEP_LINKED: the reference's main-branch neighbour structure (one atomically "
                        "built linked list per cell, no sort; simulator.cu:44-55,133-147) written for this "
                        "GPU with the library's float4 streams -- the closest thing to 'the reference's "
                        "algorithm on MI355X' that can be run (the CUDA source cannot). Same K steps, same "
                        "input; summation order is a race, results match the oracle to 1e-5 relative "
                        "(tests/test_gpu_linked.py). Not `value`."}
        tr = load_traffic(result["n_total"], args, steps=K) if world == 1 else None
        if tr:
            roof["traffic"] = tr["bytes_per_launch"]
            roof["traffic_steps"] = tr["steps"]
            roof["traffic_source"] = tr["source"]
        else:
            roof["traffic_note"] = ("no rocprofv3 PMC measurement of this exact command (n, init, sweep, math, "
                                    "steps, warmup) under profiles/traffic.json")
        if "stamps" in result:
            out["debug_stamps"] = result["stamps"]
        if "mgpu" in result:
            out["multi_gpu_driver"] = dict(result["mgpu"], host="C++ in-process driver (libsph_mgpu.so), "
                                           + ("one process per GPU, ncclSend/ncclRecv halo exchange" if world > 1
                                              else "LOOPBACK: N slabs on one GPU, device-to-device copies"))
            if world == 1:
                out["config"]["parallelism"] = f"LOOPBACK: {args.loopback_slabs} z-slabs on ONE GPU (decomposition cost study)"
        if args.cpu_steps > 0 and world == 1:  # rank 0 at N=1 only
            out["cpu_baseline"] = cpu_baseline(args.cpu_particles or n, random_init, args.cpu_steps)
            # the GPU over the SAME window (steps 1..cpu_steps of the same run), from the per-step wall
            # times of a timed loop above (-m time waits for the compute stream after every step)
            src = result if K >= args.cpu_steps else result.get("full100")
            if src and "walls" in src and len(src["walls"]) >= args.cpu_steps and not args.cpu_particles and args.mode == "time":
                w = src["walls"][args.cpu_steps - 1]
                out["cpu_baseline"]["gpu_same_window"] = {
                    "value": n * args.cpu_steps / w, "unit": "particle-steps/s",
                    "ratio": n * args.cpu_steps / w / out["cpu_baseline"]["value"],
                    "window": f"steps 1..{args.cpu_steps} of the same run, wall time of the -m time loop"}
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())

    if world > 1 or force_mgpu:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
