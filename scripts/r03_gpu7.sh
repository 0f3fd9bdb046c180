set -x
mkdir -p gpurun_out
python -m pytest tests/test_gpu_coarse.py -x -q > gpurun_out/r03_t7.log 2>&1; tail -4 gpurun_out/r03_t7.log
python scripts/r03_dbg_coarse.py 256 64 8 0.07 2 300 2>&1 | tail -3
for extra in "--coarse-cycles 1 --schur-its 8 --schur-emin 0.07 --coarse-nodes 300" "--coarse-cycles 1 --schur-its 8 --schur-emin 0.07 --coarse-nodes 1100" "--coarse-cycles 1 --schur-its 6 --schur-emin 0.1 --coarse-nodes 1100" "--coarse-cycles 1 --schur-its 10 --schur-emin 0.05 --coarse-nodes 1100" "--coarse-cycles 2 --schur-its 6 --schur-emin 0.1 --coarse-nodes 1100" "--coarse-cycles 2 --schur-its 8 --schur-emin 0.07 --coarse-nodes 300"; do
  KKT_VERBOSE=1 python bench.py --no-cpu-baseline --no-config4 --steps 20 --warmup 5 --schur-emax 2.1 $extra 2> gpurun_out/r03_b7.err | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('CFG2', '$extra', '| its/s', round(d['value'],1), 'pc', round(d['config']['pc_apply_ms'],3), 'tts', d['config']['time_to_solution']['iterations'], round(d['config']['time_to_solution']['seconds'],3), 'setup', round(d['config']['setup_s'],2), d['config']['sweeps']['form'], d['config']['sweeps']['depth'], d['config']['sweeps']['program_fallbacks'], 'sweeps ms', round(d['stages']['preconditioner_application_ms']['time_sweeps'],3))"
  grep "coarse corrections\|continued" gpurun_out/r03_b7.err | tail -2
done
