#!/bin/bash
# one call: full GPU test suite, smoke, refreshed profiles
set -e
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r01_final
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r01_final/pytest_gpu.log 2>&1 || { tail -20 gpurun_out/r01_final/pytest_gpu.log; exit 1; }
tail -2 gpurun_out/r01_final/pytest_gpu.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
bash scripts/refresh_profiles.sh
