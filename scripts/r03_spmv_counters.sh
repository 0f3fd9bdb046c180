#!/bin/bash
# Counter passes over the operator alone: fixed-width P1 form against the ragged P2 form.
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
o=gpurun_out/spmv_pmc
mkdir -p $o
i=0
# (a pass with the TA_* counters did not finish within 200 s on this pool and was dropped)
for set in "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD" \
           "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $set --kernel-trace -d $o/p$i -o c --output-format csv -- python3 scripts/r03_spmv_forms.py --cases p1,p2 --reps 5 --options capped_rows=0 > $o/p$i.log 2>&1
  echo "pass $i rc=$?"
done
python3 - <<'P'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('gpurun_out/spmv_pmc/p*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if 'kkt_spmv_rows' not in k: continue
        agg[k[:40]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in agg.items():
    print(k)
    for c, v in sorted(d.items()):
        v = v[len(v)//2:]      # the timed repetitions
        print(f'   {c:40s} {sum(v)/len(v):16.1f}  (n={len(v)})')
P
