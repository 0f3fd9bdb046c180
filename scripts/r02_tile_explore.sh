#!/bin/bash
# Round 2: tile sweep program -- parity, then bench under each program form and a sweep over the
# tile depth / workgroup size (options fall back to KKT_* variables).
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
o=gpurun_out/r02_tile
mkdir -p $o
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q > $o/parity.log 2>&1
echo "parity rc=$?"; tail -5 $o/parity.log
run() {  # name, env...
  name=$1; shift
  env "$@" timeout -k 10 200 python bench.py --no-cpu-baseline --no-config4 --steps 20 --warmup 3 > $o/$name.json 2> $o/$name.err
  echo "$name rc=$? $(python3 - <<PY
import json
try:
    d = json.load(open("$o/$name.json"))
    s = d.get("roofline_sweeps", {})
    t = d["config"]["time_to_solution"]
    print("its/s %.1f pc %.2f ms sweeps %.2f ms us/step %.3f tts %d its %.2f s" % (d["value"], d["config"]["pc_apply_ms"], s.get("total_ms", 0), s.get("us_per_phase", 0), t["iterations"], t["seconds"]))
except Exception as e:
    print("no result:", e)
PY
)"
}
for a in "$@"; do
  case $a in
    base) run dataflow KKT_PROG_MODE=dataflow ;;
    auto) run tile_auto KKT_VERBOSE=1 ;;
    d*) run tile_$a KKT_TILE_DEPTH=${a#d} KKT_VERBOSE=1 ;;
    w*) w=${a%%d*}; d=${a##*d}; run tile_$a KKT_TILE_WAVES=${w#w} KKT_TILE_DEPTH=$d KKT_VERBOSE=1 ;;
  esac
done
grep -h "tile sweep program" $o/*.err | sort | uniq
