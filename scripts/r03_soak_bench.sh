# repeated default bench runs: any fall-back of the sweep programs shows up as config.warning
for i in $(seq 1 ${1:-12}); do
  python bench.py --no-cpu-baseline --no-config4 --steps 30 --warmup 5 ${@:2} 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print($i, round(d['value'],1), d['config']['sweeps']['program_fallbacks'], d['config'].get('warning','')[-260:])"
done
