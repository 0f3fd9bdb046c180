#!/bin/bash
set -e
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
o=gpurun_out/r01_final
mkdir -p $o
timeout -k 10 300 python bench.py --workload stokes2d --steps 10 --warmup 2 > $o/bench_stokes2d.json 2> $o/stokes.err
KKT_NO_GRAPH=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $o/stokes -o s --output-format csv -- python3 bench.py --workload stokes2d --steps 3 --warmup 1 > $o/stokes_under_rocprof.json 2> $o/stokes_prof.err
echo "stokes done"; cut -c88-108 $o/bench_stokes2d.json
: > $o/other_configs.jsonl
for extra in "--scheme CN" "--mode S" "--workload heat3d --n 32 --n_t 32" "--schur-its 8 --schur-emin 0.07"; do
  timeout -k 10 300 python bench.py --no-cpu-baseline $extra >> $o/other_configs.jsonl 2>> $o/other.err
done
cut -c88-108 $o/other_configs.jsonl
KKT_PERSISTENT=0 KKT_TRANSPORT=gloo KKT_DEVICE=0 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 20 --warmup 5 > $o/bench_n2_gloo.json 2> $o/bench_n2_gloo.err || { tail -5 $o/bench_n2_gloo.err; exit 1; }
cut -c88-108 $o/bench_n2_gloo.json
