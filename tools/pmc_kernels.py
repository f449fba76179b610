#!/usr/bin/env python3
"""Per-kernel averages of rocprofv3 --pmc passes (one directory per pass).  FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE
under-reports wide coalesced reads by 2x (MI355X_MICROARCH.md, HBM section) -- the table prints raw KiB and, for FETCH_SIZE, the
x2-corrected GB.  usage: pmc_kernels.py <dir> [<dir> ...] [--match substring]"""
import csv, glob, sys, collections
dirs = [a for a in sys.argv[1:] if not a.startswith("--")]
match = sys.argv[sys.argv.index("--match") + 1] if "--match" in sys.argv else ""
if match in dirs: dirs.remove(match)
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in dirs:
    for f in glob.glob(f"{d}/**/*_counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if match in r["Kernel_Name"]:
                acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    print(k[:110])
    for c, v in sorted(cs.items()):
        m = sum(v) / len(v)
        extra = f"  = {m * 1024 * 2 / 1e9:.3f} GB read (x2)" if c == "FETCH_SIZE" else f"  = {m * 1024 / 1e9:.3f} GB written" if c == "WRITE_SIZE" else ""
        print(f"    {c:24s} launches {len(v):4d}  mean {m:16.1f}{extra}")
