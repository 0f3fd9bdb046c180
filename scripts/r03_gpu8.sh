set -x
mkdir -p gpurun_out
run() {
  KKT_VERBOSE=1 python bench.py --no-cpu-baseline --no-config4 --steps 20 --warmup 5 --schur-emax 2.1 "$@" 2> gpurun_out/r03_b8.err | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('RUN', '$*', '| its/s', round(d['value'],1), 'pc', round(d['config']['pc_apply_ms'],3), 'tts', d['config']['time_to_solution']['iterations'], round(d['config']['time_to_solution']['seconds'],3), 'setup', round(d['config']['setup_s'],2), d['config']['sweeps']['form'], d['config']['sweeps']['depth'], d['config']['sweeps']['program_fallbacks'], 'sweeps ms', round(d['stages']['preconditioner_application_ms']['time_sweeps'],3), 'batched', round(d['stages']['preconditioner_application_ms']['batched_steps'],3))"
  grep "coarse corrections\|continued" gpurun_out/r03_b8.err | tail -2
}
for v in "10 0.05" "12 0.035" "10 0.07" "8 0.05" "12 0.05" "14 0.03"; do set -- $v; run --coarse-cycles 1 --schur-its $1 --schur-emin $2; done
run --coarse-cycles 1 --schur-its 10 --schur-emin 0.05 --coarse-cell 6
run --coarse-cycles 1 --schur-its 8 --schur-emin 0.07 --scheme CN
run --coarse-cycles 1 --schur-its 10 --schur-emin 0.05 --scheme CN
# config 4
run --workload heat3d --n 64 --n_t 128 --steps 10 --warmup 2 --spmv-reps 10 --schur-its 34 --schur-emin 7.44e-3
for v in "8 0.07" "6 0.1" "10 0.05"; do set -- $v; run --workload heat3d --n 64 --n_t 128 --steps 10 --warmup 2 --spmv-reps 10 --coarse-cycles 1 --schur-its $1 --schur-emin $2; done
