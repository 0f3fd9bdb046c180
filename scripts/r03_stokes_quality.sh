#!/bin/bash
# Time to solution of the Stokes leg (128 x 128 P2-P1, n_t = 32) for several forms of the
# sub-solves: which settings converge, in how many outer iterations, at what cost.
cd "$GRAFT_REPO_ROOT"
o=gpurun_out/st_quality
mkdir -p $o
: > $o/lines.jsonl
run() {
  echo "== $*" >> $o/err.log
  # (base flags: what bench.py's defaults were when the first three groups ran -- plain K_p
  # polynomial, estimated unless --kp-its is given, lower bound 2e-3 unless --kp-emin is given)
  timeout -k 10 330 python bench.py --workload stokes2d --steps 10 --warmup 2 --kp-coarse-cycles 0 --kp-its -1 --kp-emin 0.002 "$@" >> $o/lines.jsonl 2>> $o/err.log
  echo "[$*] rc=$?"
}
if [ "$1" = "fourth" ]; then      # two-grid form of the K_p solve
run --kp-coarse-cycles 2 --kp-its 8 --kp-emin 0.07
run --kp-coarse-cycles 3 --kp-its 8 --kp-emin 0.07
run --kp-coarse-cycles 2 --kp-its 12 --kp-emin 0.05
run --kp-coarse-cycles 3 --kp-its 16 --kp-emin 0.05
run --kp-coarse-cycles 2 --kp-its 20 --kp-emin 0.03
run --kp-coarse-cycles 2 --kp-its 12 --kp-emin 0.05 --kp-coarse-cell 2
run --kp-coarse-cycles 4 --kp-its 12 --kp-emin 0.05
elif [ "$1" = "third" ]; then
run --coarse-cycles 3 --coarse-cell 8 --kp-its 300 --kp-emin 0.0005
run --coarse-cycles 2 --coarse-cell 8 --kp-its 600 --kp-emin 0.0002
run --coarse-cycles 2 --coarse-cell 8 --schur-its 12 --schur-emin 0.07 --kp-its 300 --kp-emin 0.0005
run --coarse-cycles 2 --coarse-cell 8 --kp-its 400 --kp-emin 0.0003
elif [ "$1" = "second" ]; then
run --coarse-cycles 1 --coarse-cell 8 --kp-its 160
run --coarse-cycles 2 --coarse-cell 8 --kp-its 300 --kp-emin 0.0005
run --coarse-cycles 2 --coarse-cell 8 --kp-its 80
run --coarse-cycles 2 --coarse-cell 16 --kp-its 160
run --coarse-cycles 3 --coarse-cell 8 --kp-its 160
else
run
run --schur-its 80 --schur-emin 0.002 --kp-its 80 --kp-coarse-cycles 0
run --schur-its 160 --schur-emin 0.002 --kp-its 160
run --coarse-cycles 2 --coarse-cell 8
run --coarse-cycles 3 --coarse-cell 8
run --coarse-cycles 2 --coarse-cell 8 --schur-its 12 --schur-emin 0.04
run --coarse-cycles 2 --coarse-cell 16 --schur-its 16 --schur-emin 0.02
run --coarse-cycles 2 --coarse-cell 8 --kp-its 160
fi
python3 - <<'P'
import json
for l in open('gpurun_out/st_quality/lines.jsonl'):
    try: d = json.loads(l)
    except Exception: continue
    c = d['config']; t = c.get('time_to_solution') or {}
    print(round(d['value'], 1), 'its/s  pc', round(c['pc_apply_ms'], 1), 'ms |', c['preconditioner'][21:150])
    print('      TTS', t.get('converged'), t.get('iterations'), round(t.get('seconds', 0), 2), 's  verr', t.get('velocity_error'), 'perr', t.get('pressure_error'))
P
