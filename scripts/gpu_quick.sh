#!/bin/bash
# Fast GPU loop: small parity subset, bench, kernel-trace stats, HBM-traffic PMC passes.
# usage: scripts/gpu_quick.sh TAG
set -o pipefail
TAG=$1; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout -k 10 200 python -m pytest tests/test_gpu_parity.py tests/test_slab.py -q -m gpu -p no:cacheprovider -x -k "pressure or skewed or coincident or loopback or click" > $OUT/pytest.log 2>&1
echo "pytest exit $?"; tail -n 3 $OUT/pytest.log
timeout -k 10 200 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench exit $?"
python - <<PY
import json
d=json.load(open("$OUT/bench.json")); k=d["kernel_ms_per_step"]; r=d["roofline"]
print("value %.3e ms/step %.3f | density %.3f force %.3f sort %.3f gather %.3f hash %.3f d2h %.3f | valu_frac d %.3f f %.3f | cpu %.3e x%d cores" % (d["value"], d["ms_per_step"], k["density"], k["force_integrate"], k["sort"], k["gather_cells"], k["hash"], k["readback_d2h"], r.get("valu_frac",0), r.get("force_valu_frac",0), d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"]))
PY
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o bench -- python3 $GRAFT_REPO_ROOT/bench.py --cpu-steps 0 > $OUT/prof_bench.json 2> $OUT/prof.err; echo "rocprof exit $?"
head -12 $OUT/prof/bench_kernel_stats.csv | cut -c1-150
rm -f $OUT/prof/bench_kernel_trace.csv
i=0
for ctrs in "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $ctrs --output-format csv -d $OUT/pmc$i -o pmc -- python3 $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 1 --cpu-steps 0 > $OUT/pmc$i.json 2> $OUT/pmc$i.err; echo "pmc pass $i exit $?"
done
cd $GRAFT_REPO_ROOT
python3 scripts/pmc_summary.py $OUT > $OUT/pmc_summary.csv; grep -E "density|force" $OUT/pmc_summary.csv
find $OUT -name "*counter_collection.csv" -size +4M -delete
