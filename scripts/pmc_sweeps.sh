#!/bin/bash
# HBM-side traffic of the persistent sweep program (pc_row_program_g) of the default bench:
# FETCH_SIZE and WRITE_SIZE in their own rocprofv3 --pmc passes (MI355X_MICROARCH.md, HBM).
set -e
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp KKT_NO_GRAPH=1
o=gpurun_out/pmc_sweeps
mkdir -p $o
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 400 rocprofv3 --pmc $c --kernel-trace -d $o/$c -o p --output-format csv -- python3 bench.py --no-cpu-baseline --steps 3 --warmup 1 > $o/$c.json 2> $o/$c.err
  echo "$c pass done"
done
python3 scripts/pmc_sweeps_summary.py $o
