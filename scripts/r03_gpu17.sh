python -m pytest tests/test_gpu_coarse.py -q 2>&1 | tail -2
for v in "8 2 8 0.07" "10 2 8 0.07" "10 2 10 0.05" "12 2 10 0.05" "12 2 12 0.04"; do set -- $v
  KKT_VERBOSE=1 python bench.py --no-cpu-baseline --no-config4 --steps 20 --warmup 5 --coarse-cell $1 --coarse-cycles $2 --schur-its $3 --schur-emin $4 2>gpurun_out/r03_b17.err | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cell $1 $2x$3 $4', '| its/s', round(d['value'],1), 'pc', round(d['config']['pc_apply_ms'],3), 'sweeps', round(d['stages']['preconditioner_application_ms']['time_sweeps'],3), 'tts', d['config']['time_to_solution']['iterations'], round(d['config']['time_to_solution']['seconds'],3), d['config']['sweeps']['depth'], d['config']['sweeps']['program_fallbacks'])"
done
CASES=40 SEED=7 timeout -k 10 400 python scripts/fuzz_coarse.py 2>&1 | tail -2
