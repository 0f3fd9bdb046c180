"""Summarise the --pmc passes of scripts/pmc_sweeps.sh: per-launch FETCH_SIZE / WRITE_SIZE
(KiB in rocprofv3's csv) of the sweep program and of the KKT SpMV, as JSON."""
import csv, glob, json, os, sys

o = sys.argv[1]
out = {}
for counter in ("FETCH_SIZE", "WRITE_SIZE"):
    files = glob.glob(os.path.join(o, counter, "**", "*counter_collection.csv"), recursive=True)
    per = {}
    for f in files:
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] != counter:
                continue
            name = row["Kernel_Name"]
            key = ("pc_row_program_g" if "pc_row_program_g" in name else
                   "kkt_spmv_rows" if "kkt_spmv_rows" in name else None)
            if key:
                per.setdefault(key, []).append(float(row["Counter_Value"]))
    for key, vals in per.items():
        out.setdefault(key, {})[counter] = {
            "launches": len(vals), "mean_KiB": sum(vals) / len(vals),
            "min_KiB": min(vals), "max_KiB": max(vals)}
json.dump(out, open(os.path.join(o, "summary.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
