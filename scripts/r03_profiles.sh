#!/bin/bash
# Round 3 profiles (scripts/r03_collect_profiles.py copies them from gpurun_out/r03_prof into
# profiles/r03).  rocprofv3 runs the timed path now: the captured preconditioner applications hold
# kernel nodes only (round 2's memset nodes next to the persistent kernels crashed rocprofv3 of
# ROCm 7.2 inside hipGraphLaunch).  PMC passes are separate runs (FETCH_SIZE / WRITE_SIZE).
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
o=gpurun_out/r03_prof
mkdir -p $o
MODE=$1      # "", "pmc-only" or "nopmc"
if [ "$1" != "pmc-only" ]; then
timeout -k 10 900 python bench.py > $o/bench_r03.json 2> $o/bench_r03.err
echo "bench rc=$?"
# (KKT_NO_GRAPH=1: the same kernels as plain launches.  With the captured applications rocprofv3 of
# ROCm 7.2 ran until the coarse exchange got its second pair of granule buffers and has crashed
# inside hipGraphLaunch since -- twice out of two, in its own frames below kkt::SchurPC::run)
KKT_NO_GRAPH=1 timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $o/heat -o h --output-format csv -- python3 bench.py --no-cpu-baseline --no-config4 > $o/bench_under_rocprof.json 2> $o/heat.err
echo "kernel stats rc=$?"
fi
pmc() {   # name, then the bench arguments (each pass bounded: a hung pass must not take the call)
  [ "$MODE" = "nopmc" ] && return 0
  name=$1; shift
  for c in FETCH_SIZE WRITE_SIZE; do
    KKT_NO_GRAPH=1 timeout -k 10 240 rocprofv3 --pmc $c --kernel-trace -d $o/${name}_$c -o p --output-format csv -- python3 bench.py --no-cpu-baseline --no-config4 "$@" > $o/${name}_$c.json 2> $o/${name}_$c.err
    echo "$name $c rc=$?"
  done
}
# the operator alone (no preconditioner is built: rocprofv3 --pmc of ROCm 7.2 segfaulted within a
# second of the run with the two-grid set-up's thousands of small launches in it)
pmc heat --only-spmv --spmv-reps 20
pmc cn --only-spmv --spmv-reps 20 --scheme CN
pmc cfg4 --only-spmv --spmv-reps 10 --workload heat3d --n 64 --n_t 128
pmc stokes --only-spmv --spmv-reps 10 --workload stokes2d
# the sweep program of the plain (round-2) preconditioner and of the two-grid one
pmc sweep_plain --coarse-cycles 0 --steps 2 --warmup 1
pmc sweep_twogrid --steps 2 --warmup 1
: > $o/other_configs.jsonl
for extra in "--coarse-cycles 0" "--scheme CN" "--scheme CN --coarse-cycles 0" "--mode S" "--coarse-cycles 1 --schur-its 10" "--n 512 --steps 10 --warmup 2 --coarse-cell 16"; do
  timeout -k 10 300 python bench.py --no-cpu-baseline --no-config4 $extra >> $o/other_configs.jsonl 2>> $o/other.err
  echo "variant [$extra] rc=$?"
done
timeout -k 10 400 python bench.py --workload stokes2d --steps 10 --warmup 2 > $o/bench_stokes2d.json 2> $o/stokes.err
echo "stokes rc=$?"
# the round-2 preconditioner of this leg (plain polynomials of 40 sweeps): faster per iteration, does not converge
timeout -k 10 400 python bench.py --workload stokes2d --steps 10 --warmup 2 --coarse-cycles 0 --schur-its 40 --schur-emin 0.002 --kp-coarse-cycles 0 --kp-its 40 --kp-emin 0.002 --tts-max-it 200 > $o/bench_stokes2d_round2_pc.json 2>> $o/stokes.err
echo "stokes round-2 pc rc=$?"
# plain 600-sweep pressure-Laplacian polynomial (the default before its two-grid form)
timeout -k 10 400 python bench.py --workload stokes2d --steps 10 --warmup 2 --kp-coarse-cycles 0 > $o/bench_stokes2d_plain_kp.json 2>> $o/stokes.err
echo "stokes plain kp rc=$?"
python3 scripts/r03_tts_quality.py > $o/tts_quality.txt 2>&1
ls $o
