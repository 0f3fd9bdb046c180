set -x
mkdir -p gpurun_out
runs() {
  KKT_VERBOSE=1 python bench.py --workload stokes2d --steps 5 --warmup 2 "$@" 2> gpurun_out/r03_b10.err | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('STOKES', '$*', '| its/s', round(d['value'],2), 'pc', round(d['config']['pc_apply_ms'],2), 'op', round(d['config']['kkt_apply_ms'],3), d['config']['sweeps'], {k: (round(v,2) if isinstance(v,float) else v) for k,v in d['stages'].get('velocity_preconditioner_application_ms',{}).items() if k!='note'}, 'vel op', d['stages'].get('velocity_operator_apply_ms'))"
  grep "continued\|coarse corrections" gpurun_out/r03_b10.err | tail -2
}
runs
runs --coarse-cycles 1 --schur-its 10 --schur-emin 0.07
runs --coarse-cycles 1 --schur-its 8 --schur-emin 0.07 --coarse-cell 16
python bench.py --no-cpu-baseline --no-config4 --steps 20 --warmup 5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('DEFAULT', round(d['value'],1), d['config']['preconditioner'], d['config']['time_to_solution'], d['stages']['krylov_iteration_ms'])"
KKT_DEVICE=0 python bench.py --gpus 2 --no-cpu-baseline --no-config4 --steps 6 --warmup 2 2>gpurun_out/r03_b10b.err | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('TWO RANKS', round(d['value'],1), d['config']['transport'], d['config']['time_to_solution'], d['config']['sweeps'])"
tail -3 gpurun_out/r03_b10b.err
