# round-3 GPU session 3
set -x
mkdir -p gpurun_out
python -m pytest tests/test_gpu_edge_cases.py -x -q -k "resident or timeout" > gpurun_out/r03_t3.log 2>&1; tail -5 gpurun_out/r03_t3.log
for lib in control_amd/libkkt_serial.so control_amd/libkkt.so; do
  KKT_LIB=$PWD/$lib python bench.py --workload stokes2d --steps 5 --warmup 2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('STOKES', '$lib', d['value'], d['config']['kkt_apply_ms'], d['config']['pc_apply_ms'], d['roofline']['frac'])"
done
# MMS at the reference's parameters (n_t = 30, rtol = atol = 1e-7, 200 iterations, non-linear 1e-6)
for n in 8 16 32; do
python scripts/ns_reference_nu.py --case mms --n $n --n_t 30 --max-it 200 --rtol 1e-7 --atol 1e-7 --nl-tol 1e-6 2>&1 | tail -2
done
python scripts/ns_reference_nu.py --case mms --n 16 --n_t 30 --max-it 200 --rtol 1e-7 --atol 1e-7 --nl-tol 1e-6 --cn 2>&1 | tail -2
python scripts/ns_reference_nu.py --case cavity --n 32 --max-it 300 2>&1 | tail -2
# rocprofv3 with the preconditioner's hipGraphs (kernel nodes only now)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r03_prof_graph -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-config4 --steps 5 --warmup 2 > $GRAFT_REPO_ROOT/gpurun_out/r03_prof_graph.json 2> $GRAFT_REPO_ROOT/gpurun_out/r03_prof_graph.err; echo "rocprof with graphs rc $?"
tail -3 $GRAFT_REPO_ROOT/gpurun_out/r03_prof_graph.err
