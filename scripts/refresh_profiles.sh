#!/bin/bash
# refresh gpurun_out/r01_final: default bench, kernel stats of the same command, Stokes bench + stats
set -e
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
o=gpurun_out/r01_final
mkdir -p $o
timeout -k 10 500 python bench.py > $o/bench_r01.json 2> $o/bench_r01.err
echo "bench done"; tail -c 600 $o/bench_r01.json
KKT_NO_GRAPH=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $o/heat -o h --output-format csv -- python3 bench.py --no-cpu-baseline > $o/bench_under_rocprof.json 2> $o/heat.err
echo "heat profile done"
timeout -k 10 300 python bench.py --workload stokes2d --steps 20 --warmup 3 > $o/bench_stokes2d.json 2> $o/stokes.err
KKT_NO_GRAPH=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $o/stokes -o s --output-format csv -- python3 bench.py --workload stokes2d --steps 5 --warmup 1 > $o/stokes_under_rocprof.json 2> $o/stokes_prof.err
echo "stokes profile done"
: > $o/other_configs.jsonl
for extra in "--scheme CN" "--mode S" "--workload heat3d --n 32 --n_t 32"; do
  timeout -k 10 300 python bench.py --no-cpu-baseline $extra >> $o/other_configs.jsonl 2>> $o/other.err
done
cut -c88-108 $o/other_configs.jsonl
ls -R $o | head -30
