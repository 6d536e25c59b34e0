"""The `./sph` command line (cudafluidsimulator_amd/csrc/main.cpp) keeps the reference's
interface (main.cpp:12-55): -n / -i random|grid / -m free|time / -?, rejection of bad
values with the usage text and exit status 1, and after `-m time` the five-line table
of times.h (main.cpp:69-76)."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SPH = os.path.join(ROOT, "cudafluidsimulator_amd", "sph")

USAGE = ("Program Options:\n"
         "  -n  <NUM_PARTICLES>    Number of particles to simulate\n"
         "  -i  <random/grid>      Initialization mode: random or grid\n"
         "  -m  <free/time>        Execution mode: free or timed\n"
         "  -?                     This message\n")


def run(*args, env=None):
    if not os.path.exists(SPH):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "cudafluidsimulator_amd", "csrc")],
                              stdout=subprocess.DEVNULL)
    e = dict(os.environ)
    e.update(env or {})
    return subprocess.run([SPH, *args], capture_output=True, text=True, timeout=120, env=e)


@pytest.mark.parametrize("args,first", [(("-i", "bogus"), "Invalid argument for option -i: bogus\n"),
                                        (("-m", "fast"), "Invalid argument for option -m: fast\n"),
                                        (("-?",), "")])
def test_bad_arguments_print_usage_and_exit_1(args, first):
    r = run(*args)
    assert r.returncode == 1
    assert r.stdout == first + USAGE


def test_without_a_gpu_the_cli_fails_loudly():
    """No CPU fallback: where no device is visible the product refuses to run."""
    r = run("-n", "100", "-i", "grid", "-m", "time", env={"HIP_VISIBLE_DEVICES": "-1",
                                                          "ROCR_VISIBLE_DEVICES": "-1"})
    assert r.returncode != 0
    assert "Grid construction" not in r.stdout
    assert r.stderr.strip() or r.stdout.strip()


@pytest.mark.gpu
def test_timed_run_prints_the_reference_table():
    r = run("-n", "8192", "-i", "grid", "-m", "time")
    assert r.returncode == 0, r.stderr
    lines = r.stdout.splitlines()
    i = next(k for k, l in enumerate(lines) if l.startswith("Operation"))
    head, rule, grid, sph_, xfer = lines[i:i + 5]
    assert head.split() == ["Operation", "Per", "frame", "Total"]
    assert re.fullmatch(r"-+", rule)
    for row, name in ((grid, "Grid construction"), (sph_, "SPH update"), (xfer, "Data transfer")):
        assert row.startswith(name)
        per_frame, total = re.findall(r"\d+\.\d{5}", row)      # "%.5f" like times.h:23-33
        assert float(total) >= float(per_frame) >= 0.0
    assert float(re.findall(r"\d+\.\d{5}", sph_)[1]) > 0.0


@pytest.mark.gpu
def test_free_mode_runs_headless():
    r = run("-n", "4096", "-i", "random", "-m", "free", env={"SPH_FREE_FRAMES": "5"})
    assert r.returncode == 0, r.stderr


def _sha_of(stdout):
    m = re.search(r"^positions_sha256 ([0-9a-f]{64})$", stdout, re.M)
    assert m, stdout
    return m.group(1)


@pytest.mark.gpu
def test_cli_result_equals_the_c_abi_path_and_the_oracle():
    """The C++ `Simulator` behind ./sph (the drop-in surface itself) is checked
    numerically, not just for its table: sha256 of getPosition() after the 100 steps
    of `-m time` equals the ctypes path's and the CPU oracle's on the same input."""
    import hashlib

    import numpy as np

    import cudafluidsimulator_amd as sph
    from oracle import oracle as O

    n = 8192
    r = run("-n", str(n), "-i", "grid", "-m", "time", env={"SPH_PRINT_SHA256": "1"})
    assert r.returncode == 0, r.stderr
    got = _sha_of(r.stdout)
    sim = sph.Simulator(sph.default_settings(n, False))
    sim.setup()
    t = sph.Times()
    for _ in range(100):
        sim.simulateAndTime(t)
    via_abi = hashlib.sha256(np.ascontiguousarray(np.array(sim.getPosition())).tobytes()).hexdigest()
    sim.close()
    ref = O.OracleSim(n, False)
    ref.setup()
    ref.step(100)
    want = hashlib.sha256(np.ascontiguousarray(ref.download()["pos"]).tobytes()).hexdigest()
    ref.close()
    assert got == via_abi == want


@pytest.mark.gpu
def test_cli_headline_run_matches_the_oracles_step_100_checksum():
    """`./sph -n 4194304 -i random -m time` -- the benchmark command line itself --
    ends on the positions the CPU oracle computed (tests/golden, step 100)."""
    import json
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "random4194304_sha256.json")))
    r = run("-n", str(gold["n"]), "-i", "random", "-m", "time", env={"SPH_PRINT_SHA256": "1"})
    assert r.returncode == 0, r.stderr
    assert _sha_of(r.stdout) == gold["steps"]["100"]["pos_sha256"]


@pytest.mark.gpu
def test_cli_free_mode_with_click_matches_the_oracle():
    """-m free (simulate() + the mouseClicked handshake, simulator.cu:482-489): 6 frames
    with a click after frame 3, against the oracle doing the same."""
    import hashlib

    import numpy as np

    from oracle import oracle as O

    n = 20000
    r = run("-n", str(n), "-i", "random", "-m", "free",
            env={"SPH_FREE_FRAMES": "6", "SPH_FREE_CLICK": "1", "SPH_PRINT_SHA256": "1"})
    assert r.returncode == 0, r.stderr
    ref = O.OracleSim(n, True)
    ref.setup()
    for f in range(6):
        ref.step()
        if f == 3:  # headless.cpp sets mouseClicked before frame frames/2; simulate() applies it after the step
            ref.click(400, 300)
    want = hashlib.sha256(np.ascontiguousarray(ref.download()["pos"]).tobytes()).hexdigest()
    ref.close()
    assert _sha_of(r.stdout) == want


@pytest.mark.gpu
def test_cli_multi_gpu_mode_equals_single_gpu():
    """SPH_GPUS=N routes the same Simulator class through the in-process multi-GPU
    driver (include/sph_mgpu.h); with the loopback transport the N slabs share the one
    GPU of the test box.  Same sha256 as the single-GPU run, table still printed."""
    args = ("-n", "262144", "-i", "random", "-m", "time")
    one = run(*args, env={"SPH_PRINT_SHA256": "1"})
    assert one.returncode == 0, one.stderr
    four = run(*args, env={"SPH_PRINT_SHA256": "1", "SPH_GPUS": "4", "SPH_TRANSPORT": "loopback"})
    assert four.returncode == 0, four.stderr
    assert "Grid construction" in four.stdout
    assert _sha_of(four.stdout) == _sha_of(one.stdout)
