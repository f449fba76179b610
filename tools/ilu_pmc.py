#!/usr/bin/env python3
"""HBM traffic of the wavefront triangular-solve kernels from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of
tools/ilu_only.py.  Counters are KiB; FETCH_SIZE is reported raw and with the x2 gfx950 correction for wide reads
(MI355X_MICROARCH.md, HBM section) -- the loader's 16-byte-per-lane reads are not "wide coalesced", so the truth lies between.
usage: ilu_pmc.py <fetch_dir> <write_dir> <grid>"""
import csv, glob, sys, collections
def means(d, counter):
    f = glob.glob(f"{d}/**/*_counter_collection.csv", recursive=True)[0]
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter: acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}
fm, wm = means(sys.argv[1], "FETCH_SIZE"), means(sys.argv[2], "WRITE_SIZE")
n = int(sys.argv[3]) ** 3
for k in fm:
    if "tri_wave" in k or "tri_grid" in k or "fill" in k:
        fwd = "<true>" in k
        alg_r = (32 if fwd else 40) * n if "fill" not in k else 0
        alg_w = 8 * n
        print(f"{k[:60]:60s} read {fm[k] * 1024 / 1e6:9.1f} MB raw (x2: {fm[k] * 2048 / 1e6:9.1f}; algorithmic {alg_r / 1e6:7.1f})   write {wm.get(k, 0) * 1024 / 1e6:9.1f} MB (algorithmic {alg_w / 1e6:7.1f})")
