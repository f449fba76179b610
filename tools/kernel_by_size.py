#!/usr/bin/env python3
"""Per (kernel, grid size) averages from a rocprofv3 --kernel-trace CSV: bench.py runs the same kernels on two problem sizes
(512^3 and 256^3) in one process, and rocprofv3 --stats averages over both; this keeps them apart.
usage: kernel_by_size.py <dir with *_kernel_trace.csv> [out.csv]"""
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*_kernel_trace.csv", recursive=True)[0]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    g = int(r.get("Grid_Size", r.get("Grid_Size_X", 0)) or 0)
    wg = int(r.get("Workgroup_Size", r.get("Workgroup_Size_X", 0)) or 0)
    acc[(r["Kernel_Name"], g, wg)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
rows = sorted(((sum(v), k, v) for k, v in acc.items()), reverse=True)
out = open(sys.argv[2], "w") if len(sys.argv) > 2 else sys.stdout
w = csv.writer(out)
w.writerow(["Kernel_Name", "Grid_Size", "Workgroup_Size", "Calls", "TotalDurationUs", "AverageUs", "MinUs", "MaxUs"])
for tot, (name, g, wg), v in rows:
    w.writerow([name[:160], g, wg, len(v), f"{tot:.1f}", f"{tot / len(v):.2f}", f"{min(v):.2f}", f"{max(v):.2f}"])
