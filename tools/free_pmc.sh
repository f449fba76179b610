#!/bin/bash
# instruction counters of the narrow-level run kernels on the random band matrix (one --pmc pass per group), summarised per kernel name
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/free_pmc; mkdir -p $O
i=0
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES" "SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $grp --output-format csv -d $O/g$i -o c -- python3 $R/tools/band_apply.py ${1:-2000000} > $O/g$i.log 2>&1 || { tail -3 $O/g$i.log; exit 1; }
done
python3 - "$O" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob(sys.argv[1] + "/g*/**/*counter_collection.csv", recursive=True):
    seen = set()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:60]
        if "tri_run" not in k: continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] not in seen: seen.add(r["Counter_Name"])
        cnt[(k, r["Counter_Name"])] += 1
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()): print(f"   {c:24s} {v / cnt[(k, c)]:16.0f} per launch  ({cnt[(k, c)]} launches)")
PY
