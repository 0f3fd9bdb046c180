set -x
mkdir -p gpurun_out
run() {
  KKT_VERBOSE=1 python bench.py --no-cpu-baseline --no-config4 --steps 20 --warmup 5 --schur-emax 2.1 "$@" 2> gpurun_out/r03_b9.err | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('RUN', '$*', '| its/s', round(d['value'],1), 'pc', round(d['config']['pc_apply_ms'],3), 'tts', d['config']['time_to_solution']['iterations'], round(d['config']['time_to_solution']['seconds'],3), 'setup', round(d['config']['setup_s'],2), d['config']['sweeps']['form'], d['config']['sweeps']['depth'], d['config']['sweeps']['program_fallbacks'], 'sweeps ms', round(d['stages']['preconditioner_application_ms']['time_sweeps'],3), 'batched', round(d['stages']['preconditioner_application_ms']['batched_steps'],3))"
  grep "continued" gpurun_out/r03_b9.err | tail -2
}
python -m pytest tests/test_gpu_coarse.py -x -q 2>&1 | tail -2
for v in "8 0.07" "6 0.1" "10 0.07" "12 0.05"; do set -- $v; run --workload heat3d --n 64 --n_t 128 --steps 10 --warmup 2 --spmv-reps 10 --coarse-cycles 1 --schur-its $1 --schur-emin $2; done
for v in "2 10 0.07" "3 8 0.07" "2 16 0.03" "1 24 0.02" "3 10 0.07"; do set -- $v; run --scheme CN --coarse-cycles $1 --schur-its $2 --schur-emin $3; done
