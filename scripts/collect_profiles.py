#!/usr/bin/env python3
"""Copy one round's evidence from gpurun_out/<TAG_K20>, <TAG_K100> (scripts/gpu_profile_round.sh) into profiles/
under the round's prefix and refresh profiles/traffic.json (older entries of the same command are kept, marked
current: false).  usage: scripts/collect_profiles.py r03 gpurun_out/r03_prof_k20 gpurun_out/r03_prof_k100"""
import csv
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd = sys.argv[1]
dirs = {20: sys.argv[2], 100: sys.argv[3]}
tpath = os.path.join(ROOT, "profiles", "traffic.json")
traffic = json.load(open(tpath))
for K, d in dirs.items():
    tag = f"{rnd}_bench_k{K}_w5"
    shutil.copy(os.path.join(d, "bench.json"), os.path.join(ROOT, "profiles", f"{tag}.json"))
    shutil.copy(os.path.join(d, "prof_bench.json"), os.path.join(ROOT, "profiles", f"{tag}_under_rocprof.json"))
    shutil.copy(os.path.join(d, "kernel_stats_timed.csv"), os.path.join(ROOT, "profiles", f"{rnd}_kernel_stats_timed_k{K}_w5.csv"))
    ks = os.path.join(d, "prof", "bench_kernel_stats.csv")
    if os.path.exists(ks):
        shutil.copy(ks, os.path.join(ROOT, "profiles", f"{rnd}_kernel_stats_k{K}_w5.csv"))
    shutil.copy(os.path.join(d, "pmc_summary.csv"), os.path.join(ROOT, "profiles", f"{rnd}_pmc_hbm_k{K}_w5.csv"))
    rows = {(r["kernel"], r["counter"]): r for r in csv.DictReader(open(os.path.join(d, "pmc_summary.csv")))}
    kern = next(k for k, _ in rows if "k_density_mask_lds<false" in k)
    col = f"avg_last_{K}_dispatches"
    fetch = float(rows[kern, "FETCH_SIZE"][col])
    write = float(rows[kern, "WRITE_SIZE"][col])
    for e in traffic:
        if e.get("sweep") == "list" and e.get("steps") == K and e.get("n") == 4194304:
            e["current"] = False
    traffic.append({
        "n": 4194304, "init": "random", "sweep": "list", "math": "strict", "gpus": 1, "steps": K, "warmup": 5,
        "round": int(rnd[1:]), "current": True, "kernel": kern.strip('"'),
        "FETCH_SIZE_KB_per_launch": fetch, "WRITE_SIZE_KB_per_launch": write,
        "bytes_per_launch": int((2 * fetch + write) * 1024),
        "source": f"profiles/{rnd}_pmc_hbm_k{K}_w5.csv: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate runs of "
                  f"`bench.py --steps {K} --warmup 5 --no-extra-legs`), averaged over the {K} timed launches; bytes = "
                  "(2 x FETCH_SIZE + WRITE_SIZE) x 1024 (gfx950: FETCH_SIZE counts half the bytes of 16-B/lane streaming "
                  "reads, MI355X_MICROARCH.md HBM section). The writes are the hit stream handed to the force sweep."})
json.dump(traffic, open(tpath, "w"), indent=1)
print("profiles/ refreshed:", [e["bytes_per_launch"] for e in traffic if e.get("current")])
