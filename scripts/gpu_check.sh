#!/bin/bash
# One gpurun call: GPU parity tests, smoke, a short bench and a rocprofv3 kernel trace.
# usage: scripts/gpu_check.sh [tag]   (outputs under gpurun_out/<tag>/)
set -o pipefail
TAG=${1:-r1}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo "== pytest -m gpu" | tee $OUT/status.txt
timeout -k 10 ${PYTEST_TIMEOUT:-700} python -m pytest tests -q -m gpu -p no:cacheprovider ${PYTEST_ARGS:-} > $OUT/pytest.log 2>&1
echo "pytest exit $?" | tee -a $OUT/status.txt
tail -n 40 $OUT/pytest.log
echo "== smoke" | tee -a $OUT/status.txt
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1
echo "smoke exit $?" | tee -a $OUT/status.txt
tail -n 5 $OUT/smoke.log
