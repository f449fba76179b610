#!/bin/bash
# PMC traffic of the plain-CSR SpMV kernel (separate --pmc passes), for a few tile maps.  usage: plain_pmc.sh <grid> <tag> [ENV=VAL ...]
R=/root/repo
g=$1; tag=$2; shift 2
O=$R/gpurun_out/plain_pmc/$tag
rm -rf $O && mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export KRYST_SPMV_COMPRESS=0
for kv in "$@"; do export "$kv"; done
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/f -o f -- python3 $R/tools/spmv_only.py $g 5 1 > $O/f.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/w -o w -- python3 $R/tools/spmv_only.py $g 5 1 > $O/w.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/h -o h -- python3 $R/tools/spmv_only.py $g 5 1 > $O/h.log 2>&1 || exit 1
python3 $R/tools/pmc_traffic.py $O/f $O/w $g $O/traffic.json > /dev/null || exit 1
python3 - <<PY
import csv,glob,collections
f=glob.glob("$O/h/**/*_counter_collection.csv",recursive=True)[0]
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in acc.items():
    if "spmv" in k: print("$tag", k, {c: sum(x)/len(x) for c,x in v.items()})
PY
python3 -c "
import json; d=json.load(open('$O/traffic.json')); print('$tag', 'read', d['read_bytes_per_launch']/1e9, 'write', d['write_bytes_per_launch']/1e9, 'over', d['traffic_over_algorithmic'], 'cal', d['calibration']['factor'])"
