#!/bin/bash
# Round 2 profiles (scripts/r02_collect_profiles.py copies them from gpurun_out/r02_prof into
# profiles/r02):
#   bench line, rocprofv3 kernel stats of the same command (KKT_NO_GRAPH=1: rocprofv3 of ROCm 7.2
#   segfaults inside hipGraphLaunch of the captured preconditioner graphs; same kernels, same
#   order), PMC passes (FETCH_SIZE / WRITE_SIZE, separate runs) for the KKT SpMV and the tile
#   sweep program, bench variants.
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
o=gpurun_out/r02_prof
mkdir -p $o
timeout -k 10 600 python bench.py > $o/bench_r02.json 2> $o/bench_r02.err
echo "bench rc=$?"
KKT_NO_GRAPH=1 timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $o/heat -o h --output-format csv -- python3 bench.py --no-cpu-baseline --no-config4 > $o/bench_under_rocprof.json 2> $o/heat.err
echo "kernel stats rc=$?"
for c in FETCH_SIZE WRITE_SIZE; do
  KKT_NO_GRAPH=1 timeout -k 10 400 rocprofv3 --pmc $c --kernel-trace -d $o/$c -o p --output-format csv -- python3 bench.py --no-cpu-baseline --no-config4 --steps 3 --warmup 1 > $o/$c.json 2> $o/$c.err
  echo "$c rc=$?"
done
: > $o/other_configs.jsonl
for extra in "--scheme CN" "--mode S" "--schur-auto" "--schur-auto --scheme CN" "--n 512 --schur-auto --steps 10 --warmup 2" "--workload heat3d --n 32 --n_t 32 --schur-auto"; do
  timeout -k 10 300 python bench.py --no-cpu-baseline --no-config4 $extra >> $o/other_configs.jsonl 2>> $o/other.err
  echo "variant [$extra] rc=$?"
done
timeout -k 10 300 python bench.py --workload stokes2d --steps 10 --warmup 2 > $o/bench_stokes2d.json 2> $o/stokes.err
echo "stokes rc=$?"
python3 - <<'PY'
import csv, glob, json, os
o = "gpurun_out/r02_prof"
out = {}
for counter in ("FETCH_SIZE", "WRITE_SIZE"):
    per = {}
    for f in glob.glob(os.path.join(o, counter, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] != counter:
                continue
            name = row["Kernel_Name"]
            key = ("pc_tile_sweep" if "pc_tile_sweep" in name else
                   "kkt_spmv_rows" if "kkt_spmv_rows" in name else None)
            if key:
                per.setdefault(key, []).append(float(row["Counter_Value"]))
    for key, vals in per.items():
        out.setdefault(key, {})[counter] = {"launches": len(vals), "mean_KiB": sum(vals) / len(vals),
                                            "min_KiB": min(vals), "max_KiB": max(vals)}
json.dump(out, open(os.path.join(o, "pmc_summary.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
PY
ls $o
