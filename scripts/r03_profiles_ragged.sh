#!/bin/bash
# Partial refresh of profiles/r03 after the width-switched, XCD-ordered kernel for ragged
# structures (kkt_spmv_rows_ragged): the default bench line, the Stokes leg, the PMC passes of the
# Stokes outer operator (FETCH_SIZE / WRITE_SIZE, separate runs), kernel stats of the operator-only
# Stokes run and the forms table.  scripts/r03_collect_profiles.py copies from gpurun_out/r03_prof.
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
o=gpurun_out/r03_prof
mkdir -p $o
timeout -k 10 900 python bench.py > $o/bench_r03.json 2> $o/bench_r03.err
echo "bench rc=$?"
for c in FETCH_SIZE WRITE_SIZE; do
  KKT_NO_GRAPH=1 timeout -k 10 240 rocprofv3 --pmc $c --kernel-trace -d $o/stokes_$c -o p --output-format csv -- python3 bench.py --no-cpu-baseline --no-config4 --only-spmv --spmv-reps 10 --workload stokes2d > $o/stokes_$c.json 2> $o/stokes_$c.err
  echo "stokes $c rc=$?"
done
KKT_NO_GRAPH=1 timeout -k 10 240 rocprofv3 --kernel-trace --stats -d $o/stokes_op -o s --output-format csv -- python3 bench.py --no-cpu-baseline --no-config4 --only-spmv --spmv-reps 10 --workload stokes2d > $o/stokes_op.json 2> $o/stokes_op.err
echo "stokes op kernel stats rc=$?"
timeout -k 10 400 python bench.py --workload stokes2d --steps 10 --warmup 2 > $o/bench_stokes2d.json 2> $o/stokes.err
echo "stokes rc=$?"
# kernel stats of the whole Stokes leg (plain launches: rocprofv3 and the captured applications, see r03_profiles.sh)
KKT_NO_GRAPH=1 timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $o/stokes_leg -o s --output-format csv -- python3 bench.py --workload stokes2d --steps 5 --warmup 1 --tts-max-it 1 > $o/stokes_leg.json 2> $o/stokes_leg.err
echo "stokes leg kernel stats rc=$?"
for opt in "" "--options ragged_xcd=0" "--options ragged_switch=0"; do
  echo "forms $opt" >> $o/spmv_forms.jsonl
  timeout -k 10 300 python scripts/r03_spmv_forms.py --cases p1,q2,p2,stokes $opt >> $o/spmv_forms.jsonl 2>> $o/spmv_forms.err
done
echo "forms rc=$?"
KKT_NO_GRAPH=1 timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $o/heat -o h --output-format csv -- python3 bench.py --no-cpu-baseline --no-config4 > $o/bench_under_rocprof.json 2> $o/heat.err
echo "kernel stats rc=$?"
ls $o
