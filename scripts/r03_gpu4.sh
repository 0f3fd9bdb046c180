# round-3 GPU session 4: row-sort window A/B on the Stokes operator, then the whole GPU suite
set -x
mkdir -p gpurun_out
for s in 8 4 2 16; do
  KKT_SELL_SIGMA=$s python bench.py --workload stokes2d --steps 5 --warmup 2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('STOKES sigma', $s, d['value'], d['config']['kkt_apply_ms'], d['config']['pc_apply_ms'], d['roofline']['frac'], d['roofline']['algorithmic_bytes_per_launch'])"
done
python -m pytest tests -m gpu -x -q > gpurun_out/r03_t4.log 2>&1; tail -15 gpurun_out/r03_t4.log
