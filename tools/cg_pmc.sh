#!/bin/bash
# HBM traffic of the kernels of a CG iteration at grid^3 (default: the fused form): separate --pmc FETCH_SIZE / WRITE_SIZE passes, read side calibrated on DotOp.
G=${1:-512}
cd /tmp && export TMPDIR=/tmp
O=/root/repo/gpurun_out/cg_pmc; rm -rf $O; mkdir -p $O
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/f -o f -- python3 /root/repo/tools/cg_only.py $G 26 > $O/f.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/w -o w -- python3 /root/repo/tools/cg_only.py $G 26 > $O/w.log 2>&1 || exit 1
python3 - $O $G <<'PY'
import csv, glob, sys, collections
o, g = sys.argv[1], int(sys.argv[2]); n = g ** 3
def means(d, c):
    f = glob.glob(f"{d}/**/*_counter_collection.csv", recursive=True)[0]
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == c: acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in acc.items()}
fm, wm = means(o + "/f", "FETCH_SIZE"), means(o + "/w", "WRITE_SIZE")
dotk = [k for k in fm if "DotOp" in k][0]
cal = 2 * n * 8 / (fm[dotk][0] * 1024.0)
print(f"calibration factor {cal:.3f} (DotOp)")
for k in sorted(fm, key=lambda k: -fm[k][0]):
    if fm[k][1] < 3: continue
    rd = fm[k][0] * 1024 * cal; wr = wm.get(k, (0, 0))[0] * 1024
    print(f"{k[:90]:90s} calls {fm[k][1]:3d}  read {rd / 1e9:7.3f} GB  written {wr / 1e9:7.3f} GB  = {(rd + wr) / n:6.1f} B/row")
PY
