set -x
mkdir -p gpurun_out
python -m pytest tests/test_gpu_coarse.py -x -q > gpurun_out/r03_t6.log 2>&1; tail -4 gpurun_out/r03_t6.log
for extra in "" "--coarse-cycles 1 --schur-its 8 --schur-emin 0.07 --schur-emax 2.1" "--coarse-cycles 2 --schur-its 8 --schur-emin 0.07 --schur-emax 2.1" "--coarse-cycles 1 --schur-its 12 --schur-emin 0.035 --schur-emax 2.1" "--coarse-cycles 1 --schur-its 6 --schur-emin 0.1 --schur-emax 2.1"; do
  KKT_VERBOSE=1 python bench.py --no-cpu-baseline --no-config4 --steps 20 --warmup 5 $extra 2> gpurun_out/r03_b6.err | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('CFG2', '$extra', '| its/s', round(d['value'],1), 'pc', round(d['config']['pc_apply_ms'],3), 'tts', d['config']['time_to_solution']['iterations'], round(d['config']['time_to_solution']['seconds'],3), d['config']['sweeps'], d['stages']['preconditioner_application_ms'])"
  grep "coarse corrections\|tile sweep program:" gpurun_out/r03_b6.err | tail -2
done
