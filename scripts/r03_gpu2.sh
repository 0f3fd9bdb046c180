# round-3 GPU session 2: NS at the reference's viscosities; Stokes / CN operator A/B; new tests
set -x
mkdir -p gpurun_out
python -m pytest tests/test_gpu_sharded.py -x -q -k "timeout or self_launch or falls_back" > gpurun_out/r03_t2.log 2>&1; tail -5 gpurun_out/r03_t2.log
for lib in control_amd/libkkt_serial.so control_amd/libkkt.so; do
  KKT_LIB=$PWD/$lib python bench.py --workload stokes2d --steps 5 --warmup 2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('STOKES', '$lib', d['value'], d['config']['kkt_apply_ms'], d['config']['pc_apply_ms'], d['roofline']['frac'])"
  KKT_LIB=$PWD/$lib python bench.py --scheme CN --no-config4 --no-cpu-baseline --steps 10 --warmup 2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('CN', '$lib', d['value'], d['config']['kkt_apply_ms'], d['config']['pc_apply_ms'], d['roofline']['frac'])"
done
python scripts/ns_reference_nu.py --case cavity --n 8 --schur "(30, 0.25, 2.3, 0.5)" 2>&1 | grep -v "^\[kkt\] tile\|sweep program" | tail -4
python scripts/ns_reference_nu.py --case cavity --n 8 2>&1 | grep -v "^\[kkt\] tile\|sweep program" | tail -6
python scripts/ns_reference_nu.py --case cavity --n 8 --cn 2>&1 | tail -2
python scripts/ns_reference_nu.py --case cavity --n 16 2>&1 | tail -2
python scripts/ns_reference_nu.py --case cavity --n 32 2>&1 | tail -2
python scripts/ns_reference_nu.py --case mms --n 8 2>&1 | tail -3
python scripts/ns_reference_nu.py --case mms --n 16 2>&1 | tail -3
python scripts/ns_reference_nu.py --case mms --n 16 --cn 2>&1 | tail -3
python scripts/ns_reference_nu.py --case mms --n 8 --schur "(30, 0.25, 2.3, 0.5)" 2>&1 | tail -3
