for cell in 8 10 12; do for v in "2 8 0.07" "2 10 0.05" "2 12 0.04"; do set -- $v
  python bench.py --no-cpu-baseline --no-config4 --steps 20 --warmup 5 --coarse-cell $cell --coarse-cycles $1 --schur-its $2 --schur-emin $3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cell', $cell, '$v', '| its/s', round(d['value'],1), 'pc', round(d['config']['pc_apply_ms'],3), 'tts', d['config']['time_to_solution']['iterations'], round(d['config']['time_to_solution']['seconds'],3), d['config']['preconditioner'][60:100])"
done; done
