#!/bin/bash
# Time to solution of the Stokes leg (128 x 128 P2-P1, n_t = 32) for several forms of the
# sub-solves: which settings converge, in how many outer iterations, at what cost.
cd "$GRAFT_REPO_ROOT"
o=gpurun_out/st_quality
mkdir -p $o
: > $o/lines.jsonl
run() {
  echo "== $*" >> $o/err.log
  timeout -k 10 280 python bench.py --workload stokes2d --steps 10 --warmup 2 "$@" >> $o/lines.jsonl 2>> $o/err.log
  echo "[$*] rc=$?"
}
run
run --schur-its 40 --schur-emin 0.002 --kp-its 40
run --schur-its 80 --schur-emin 0.002 --kp-its 80
run --coarse-cycles 2 --coarse-cell 8
run --coarse-cycles 1 --coarse-cell 16
run --coarse-cycles 2 --coarse-cell 8 --kp-its 40
python3 - <<'P'
import json
for l in open('gpurun_out/st_quality/lines.jsonl'):
    try: d = json.loads(l)
    except Exception: continue
    c = d['config']; t = c.get('time_to_solution') or {}
    print(round(d['value'], 1), 'its/s  pc', round(c['pc_apply_ms'], 1), 'ms |', c['preconditioner'][21:150])
    print('      TTS', t.get('converged'), t.get('iterations'), round(t.get('seconds', 0), 2), 's  verr', t.get('velocity_error'), 'perr', t.get('pressure_error'))
P
