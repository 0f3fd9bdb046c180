#!/bin/bash
# kernel trace of the default bench (launches instead of graphs: rocprofv3 + hipGraphLaunch crashes)
set -e
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp KKT_NO_GRAPH=1
out=gpurun_out/${1:-prof_heat}
mkdir -p $out
timeout -k 10 600 rocprofv3 --kernel-trace --stats -d $out -o h --output-format csv -- python3 bench.py --no-cpu-baseline --steps 20 --warmup 5 > $out/bench.json 2> $out/bench.err
ls $out
