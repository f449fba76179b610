#!/bin/bash
# L1 (TCP) request counters of the narrow-level run kernels on the random band matrix
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/free_pmc2; mkdir -p $O
rocprofv3 -L 2>/dev/null | grep -o -E "\b(TCP|TA|TCC)_[A-Z_0-9a-z]+\b" | sort -u > $O/avail.txt
i=0
for grp in "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "TCP_TOTAL_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum" "TA_BUSY_avr TA_TA_BUSY_sum"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $grp --output-format csv -d $O/g$i -o c -- python3 $R/tools/band_apply.py ${1:-2000000} > $O/g$i.log 2>&1 || { echo "group $i failed"; tail -3 $O/g$i.log; }
done
python3 - "$O" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob(sys.argv[1] + "/g*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:60]
        if "tri_run" not in k: continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()): print(f"   {c:32s} {v / cnt[(k, c)]:16.0f} per launch  ({cnt[(k, c)]} launches)")
PY
