"""bench.py prints ONE JSON line with the driver's contract fields, the `roofline` and
`cpu_baseline` objects, and (N > 1) goes through the C++ multi-GPU driver."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(*args, env=None):
    e = dict(os.environ)
    e.update(env or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True,
                       timeout=600, env=e, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout
    return json.loads(lines[0])


def test_bench_needs_a_gpu(have_gpu):
    if have_gpu:
        pytest.skip("GPU present")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1"], capture_output=True,
                       text=True, timeout=300, cwd=ROOT)
    assert r.returncode != 0 and "GPU" in (r.stderr + r.stdout)


@pytest.mark.gpu
def test_single_gpu_line():
    d = run_bench("-n", "262144", "--steps", "6", "--warmup", "2", "--cpu-steps", "2", "--no-linked-leg")
    assert "other_configs" not in d   # the config 2 / config 4 legs ride on the default command only
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
              "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 6 and d["warmup"] == 2 and d["vs_baseline"] is None
    assert d["full_run"] is False and "first 6 of the 100 steps" in d["metric"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and 0 < r["frac"] < 1
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert r["traffic"] is None and "traffic_note" in r          # no PMC run of THIS command
    assert 0 < r["force_hit_fraction"] < 1 and r["force_valu_frac"] < 1
    # free fall: every recorded hit is an exact-zero pair and the filter drops it
    assert r["force_pair_bodies_per_launch"] <= r["force_hits_recorded_per_launch"]
    # the metric BASELINE.json names (the 100-step loop) is always measured, whatever --steps is
    f = d["full_run_100"]
    assert f["steps"] == 100 and f["value"] > 0 and f["roofline"]["bound"] == "hbm"
    assert abs(f["value"] - 262144 * 100 / (f["ms_per_step"] * 1e-3 * 100)) < 1e-3 * f["value"]
    assert set(f["kernel_ms_per_step"]) >= {"sort", "density", "force_integrate"}
    g = d["cpu_baseline"].get("gpu_same_window")
    assert g and g["ratio"] > 10
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["steps"] == 2 and "-march=native" in c["build"] and c["value"] > 0
    assert d["value"] > 10 * c["value"]


@pytest.mark.gpu
def test_multi_gpu_code_path_with_one_rank():
    """N > 1 runs one process per GPU through libsph_mgpu.so (ncclCommInitRank with a unique
    id handed out over torch.distributed).  A one-GPU box can run that path with one rank."""
    d = run_bench("-n", "262144", "--steps", "5", "--warmup", "1", "--cpu-steps", "0",
                  env={"SPH_BENCH_FORCE_MGPU": "1"})
    m = d["multi_gpu_driver"]
    assert m["host_syncs_per_step"] == 1.0 and m["slabs"] == 1 and m["owned"] == [262144]
    assert d["value"] > 0 and d["roofline"]["avg_launch_us"] > 0


@pytest.mark.gpu
def test_loopback_slab_study_line():
    d = run_bench("-n", "262144", "--steps", "5", "--warmup", "1", "--cpu-steps", "0", "--loopback-slabs", "4")
    m = d["multi_gpu_driver"]
    assert m["slabs"] == 4 and len(m["kernel_ms_per_step_per_slab"]) == 4 and sum(m["owned"]) == 262144
    assert "LOOPBACK" in d["config"]["parallelism"]
