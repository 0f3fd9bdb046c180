#!/bin/bash
# kernel-level profile of the Stokes-control workload (launches instead of graphs)
set -e
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp KKT_NO_GRAPH=1
mkdir -p gpurun_out/prof_stokes
timeout -k 10 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_stokes -o st --output-format csv -- python3 bench.py --workload stokes2d --steps 3 --warmup 1 > gpurun_out/prof_stokes/bench.json 2> gpurun_out/prof_stokes/bench.err
ls -R gpurun_out/prof_stokes | head -20
