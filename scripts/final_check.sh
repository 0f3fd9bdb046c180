#!/bin/bash
# one call: full GPU test suite, smoke, refreshed profiles
set -e
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r01_final
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r01_final/pytest_gpu.log 2>&1 || { tail -20 gpurun_out/r01_final/pytest_gpu.log; exit 1; }
tail -2 gpurun_out/r01_final/pytest_gpu.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r01_final/smoke.log 2>&1 || { tail -5 gpurun_out/r01_final/smoke.log; exit 1; }
tail -3 gpurun_out/r01_final/smoke.log
bash scripts/refresh_profiles.sh
# rehearsal of the N = 2 bench path on this one GPU (host-staged gloo transport)
KKT_PERSISTENT=0 KKT_TRANSPORT=gloo KKT_DEVICE=0 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 20 --warmup 5 > gpurun_out/r01_final/bench_n2_gloo.json 2> gpurun_out/r01_final/bench_n2_gloo.err || { tail -5 gpurun_out/r01_final/bench_n2_gloo.err; exit 1; }
cut -c1-200 gpurun_out/r01_final/bench_n2_gloo.json
