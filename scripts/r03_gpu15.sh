for d in 3 4 5 6 8; do
  KKT_TILE_DEPTH=$d python bench.py --no-cpu-baseline --no-config4 --steps 20 --warmup 5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('depth', $d, round(d['value'],1), 'pc', round(d['config']['pc_apply_ms'],3), 'sweeps', round(d['stages']['preconditioner_application_ms']['time_sweeps'],3), d['config']['sweeps']['depth'])"
done
for pd in 8 16 32 48; do
  KKT_TILE_POLL_DELAY=$pd python bench.py --no-cpu-baseline --no-config4 --steps 20 --warmup 5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('poll delay', $pd, round(d['value'],1), 'pc', round(d['config']['pc_apply_ms'],3), 'sweeps', round(d['stages']['preconditioner_application_ms']['time_sweeps'],3))"
done
