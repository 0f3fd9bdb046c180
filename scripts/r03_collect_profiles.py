#!/usr/bin/env python3
"""Copy what scripts/r03_profiles.sh left under gpurun_out/r03_prof into profiles/r03 (tracked):
bench lines, rocprofv3 kernel / domain stats (taken WITH the preconditioner's hipGraphs: the timed
path), the PMC rows of the priced kernels (FETCH_SIZE / WRITE_SIZE, separate passes) and the
per-launch HBM traffic derived from them, corrected as MI355X_MICROARCH.md prescribes: on gfx950
FETCH_SIZE reports half of the bytes of 16 B/lane streaming reads -- in kkt_spmv_rows only the
matrix-value stream is such a read.

    python scripts/r03_collect_profiles.py        (after a gpurun call of scripts/r03_profiles.sh)
"""
import csv
import glob
import json
import os
import shutil

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "r03_prof")
DST = os.path.join(ROOT, "profiles", "r03")
os.makedirs(DST, exist_ok=True)


def last_json(path):
    try:
        lines = [ln for ln in open(path).read().splitlines() if ln.strip().startswith("{")]
        return json.loads(lines[-1]) if lines else None
    except OSError:
        return None


for name in ("bench_r03.json", "bench_under_rocprof.json", "other_configs.jsonl",
             "bench_stokes2d.json", "bench_stokes2d_round2_pc.json", "bench_stokes2d_plain_kp.json",
             "tts_quality.txt"):
    if os.path.exists(os.path.join(SRC, name)):
        shutil.copy(os.path.join(SRC, name), os.path.join(DST, name))
if os.path.exists(os.path.join(SRC, "spmv_forms.jsonl")):          # scripts/r03_profiles_ragged.sh
    shutil.copy(os.path.join(SRC, "spmv_forms.jsonl"), os.path.join(DST, "spmv_forms_ragged.jsonl"))
f = glob.glob(os.path.join(SRC, "stokes_op", "**", "*kernel_stats.csv"), recursive=True)
if f:
    shutil.copy(f[0], os.path.join(DST, "stokes_operator_kernel_stats.csv"))
f = glob.glob(os.path.join(SRC, "stokes_leg", "**", "*kernel_stats.csv"), recursive=True)
if f:
    shutil.copy(f[0], os.path.join(DST, "stokes_leg_kernel_stats.csv"))
for kind in ("kernel_stats", "domain_stats"):
    f = glob.glob(os.path.join(SRC, "heat", "**", f"*{kind}.csv"), recursive=True)
    if f:
        shutil.copy(f[0], os.path.join(DST, f"bench_{kind}.csv"))


def rows_of(run, counter, key):
    out, header = [], None
    for f in glob.glob(os.path.join(SRC, f"{run}_{counter}", "**", "*counter_collection.csv"),
                       recursive=True):
        rd = csv.reader(open(f))
        header = next(rd)
        ci, ki = header.index("Counter_Name"), header.index("Kernel_Name")
        out += [r for r in rd if r[ci] == counter and key in r[ki]]
    return header, out


def mean_bytes(run, counter, key, tag):
    header, rows = rows_of(run, counter, key)
    if not rows:
        return None
    with open(os.path.join(DST, f"pmc_{tag}_{run}_{key}.csv"), "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(header)
        w.writerows(rows[:200])
    vi, ni = header.index("Counter_Value"), header.index("Kernel_Name")
    vals = [float(r[vi]) * 1024.0 for r in rows]                  # the counters are in KiB
    # launches of the kernel per operator apply: the distinct launch shapes among its rows (the
    # Stokes apply is two launches of the kernel -- two sparsity structures per velocity row)
    gi, wi = header.index("Grid_Size"), header.index("Workgroup_Size")
    shapes = len({(r[gi], r[wi]) for r in rows})
    return (sum(vals) / len(vals), len(vals), rows[0][ni].split("(")[0].replace("void kkt::", ""),
            shapes)


summary = {}
try:      # partial refreshes (scripts/r03_profiles_ragged.sh) keep the other passes' entries
    summary = json.load(open(os.path.join(DST, "pmc_summary.json")))
except (OSError, ValueError):
    pass
CORR = ("MI355X_MICROARCH.md (HBM): on gfx950 FETCH_SIZE reports 1/2 of the bytes of wide (16 B/lane) "
        "coalesced streaming reads; only the matrix-value stream of this kernel is such a read "
        "(bytes the launch must move minus 16 B per unknown and the index arrays), so half of it is "
        "added to the raw counter; index loads and x gathers (8 B/lane) are taken as reported; "
        "WRITE_SIZE is exact.")
for run, out_name in (("heat", "traffic_kkt_spmv_rows.json"), ("cn", "traffic_kkt_spmv_rows_cn.json"),
                      ("cfg4", "traffic_kkt_spmv_rows_cfg4.json"),
                      ("stokes", "traffic_stokes_outer_operator.json")):
    line = last_json(os.path.join(SRC, f"{run}_FETCH_SIZE.json"))
    f = mean_bytes(run, "FETCH_SIZE", "kkt_spmv_rows", "fetch")
    w = mean_bytes(run, "WRITE_SIZE", "kkt_spmv_rows", "write")
    if not (line and f and w):
        continue
    roof = line["roofline"]
    alg = roof["algorithmic_bytes_per_launch"]
    unknowns = line["config"]["unknowns"]
    value_stream = alg - 16 * unknowns
    workload = line["config"]["workload"]
    if run == "stokes":
        c = line["config"]
        # the key bench.py looks the Stokes traffic up by
        import re
        m = re.search(r"P2-P1 (\d+)x\d+, n_t=(\d+), .*?, (BE|CN), mode (\w)", workload)
        workload = f"stokes2d {m.group(1)} {m.group(2)} {m.group(3)} {m.group(4)}"
    # the Stokes operator apply is several launches of the kernel (bench.py prices the apply):
    # the counters are summed over the launches of one apply, not averaged over launches
    # (the figure committed before this was the average over its two launches: 0.98 x instead of 1.48 x)
    per_apply = f[3]
    f = (f[0] * per_apply, f[1] // per_apply, f[2])
    w = (w[0] * per_apply, w[1] // per_apply, w[2])
    corrected = f[0] + 0.5 * value_stream + w[0]
    d = {"workload": workload, "kernel": f[2], "kernel_launches_per_apply": per_apply,
         "fetch_bytes_raw": f[0], "write_bytes": w[0],
         "value_stream_bytes": value_stream, "hbm_bytes_per_launch_corrected": corrected,
         "bytes_the_launch_must_move": alg, "traffic_over_bytes": corrected / alg,
         "launch_ms": roof["launch_ms"], "frac_of_8TBs": roof["frac"],
         "correction": CORR, "launches": f[1]}
    json.dump(d, open(os.path.join(DST, out_name), "w"), indent=1)
    summary[run + "_spmv"] = {k: d[k] for k in ("kernel", "launch_ms", "frac_of_8TBs",
                                                "traffic_over_bytes", "launches")}
for run, out_name in (("sweep_twogrid", "traffic_pc_tile_sweep.json"),
                      ("sweep_plain", "traffic_pc_tile_sweep_plain.json")):
    line = last_json(os.path.join(SRC, f"{run}_FETCH_SIZE.json"))
    f = mean_bytes(run, "FETCH_SIZE", "pc_tile_sweep", "fetch")
    w = mean_bytes(run, "WRITE_SIZE", "pc_tile_sweep", "write")
    if not (line and f and w):
        continue
    sw = line.get("roofline_sweeps") or {}
    d = {"workload": line["config"]["workload"], "preconditioner": line["config"]["preconditioner"],
         "kernel": f[2], "phases_per_launch": (sw["phases"] // sw["launches"]) if sw else None,
         "fetch_bytes_raw": f[0], "write_bytes": w[0], "hbm_bytes_per_launch": f[0] + w[0],
         "launches": f[1],
         "note": "raw FETCH_SIZE + WRITE_SIZE of one sweep launch: granule stores are 16 B/lane "
                 "write-through stores (exact), granule polls 16 B/lane sc1 loads of single granules "
                 "and the per-level matrix-value gathers 8 B/lane loads (taken as reported)."}
    json.dump(d, open(os.path.join(DST, out_name), "w"), indent=1)
    summary[run + "_sweep"] = {k: d[k] for k in ("kernel", "phases_per_launch", "hbm_bytes_per_launch")}
json.dump(summary, open(os.path.join(DST, "pmc_summary.json"), "w"), indent=1)

# the bench line was written before these PMC passes ran: its traffic fields are re-derived from
# the passes of the SAME call (what bench.py itself does on its next run)
dst_bench = os.path.join(DST, "bench_r03.json")
line = last_json(dst_bench)
if line:
    t = os.path.join(DST, "traffic_kkt_spmv_rows.json")
    if os.path.exists(t):
        line["roofline"]["traffic"] = json.load(open(t))["hbm_bytes_per_launch_corrected"]
    t4 = os.path.join(DST, "traffic_kkt_spmv_rows_cfg4.json")
    if os.path.exists(t4) and "config4" in line and "roofline" in line["config4"]:
        line["config4"]["roofline"]["traffic"] = json.load(open(t4))["hbm_bytes_per_launch_corrected"]
    tt = os.path.join(DST, "traffic_pc_tile_sweep.json")
    sw = line.get("roofline_sweeps")
    if sw and os.path.exists(tt):
        tj = json.load(open(tt))
        if tj.get("phases_per_launch") == sw["phases"] // sw["launches"]:
            sw["traffic"] = tj["hbm_bytes_per_launch"] * sw["launches"]
            sw["hbm_GBs"] = sw["traffic"] / (sw["total_ms"] * 1e-3) / 1e9
            sw["hbm_frac"] = sw["hbm_GBs"] / 8000.0
    json.dump(line, open(dst_bench, "w"))
# the same for the Stokes leg's line (its operator traffic comes from the stokes passes)
dst_st = os.path.join(DST, "bench_stokes2d.json")
st, ts = last_json(dst_st), os.path.join(DST, "traffic_stokes_outer_operator.json")
if st and os.path.exists(ts) and st.get("n_gpus") == 1:
    st["roofline"]["traffic"] = json.load(open(ts))["hbm_bytes_per_launch_corrected"]
    json.dump(st, open(dst_st, "w"))
print(json.dumps(summary, indent=1))
print(sorted(os.listdir(DST)))
