#!/usr/bin/env python3
"""Turn two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; tools/spmv_only.py as the program) into the per-launch
HBM traffic of the SpMV kernel, following MI355X_MICROARCH.md (HBM section): the counters are in KiB; on gfx950
FETCH_SIZE under-reports wide coalesced reads by 2x, so the read side is CALIBRATED on a kernel with a known byte
count in the same process (ew_kernel<DotOp>: 2*n*8 bytes, 16 B/lane loads).  The result carries the sha of the source tree
it was measured on (kryst_amd._ffi.spmv_source_sha16: the files the SpMV kernels are built from): bench.py quotes it only beside numbers from the same sources.
usage: pmc_traffic.py <fetch_dir> <write_dir> <grid> <out.json> [merge_into.json form]"""
import csv, glob, json, os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kryst_amd._ffi import spmv_source_sha16


def means(d, counter):
    f = glob.glob(f"{d}/**/*_counter_collection.csv", recursive=True)[0]
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter:
            acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in acc.items()}


fetch, write, grid, out = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
n = grid ** 3
fm, wm = means(fetch, "FETCH_SIZE"), means(write, "WRITE_SIZE")
spmv = [k for k in fm if "spmv" in k and "<1," in k][0]
dotk = [k for k in fm if "DotOp" in k][0]
known = 2 * n * 8
cal = known / (fm[dotk][0] * 1024.0)
res = {
    "grid": grid, "kernel": spmv, "launches": fm[spmv][1], "source_sha16": spmv_source_sha16(),
    "FETCH_SIZE_KiB": fm[spmv][0], "WRITE_SIZE_KiB": wm[spmv][0],
    "calibration": {"kernel": dotk, "known_read_bytes": known, "FETCH_SIZE_KiB": fm[dotk][0], "factor": cal},
    "read_bytes_per_launch": fm[spmv][0] * 1024.0 * cal, "write_bytes_per_launch": wm[spmv][0] * 1024.0,
}
res["hbm_bytes_per_launch"] = res["read_bytes_per_launch"] + res["write_bytes_per_launch"]
nnz = 7 * n - 6 * grid * grid
res["algorithmic_bytes"] = 12 * nnz + 4 * (n + 1) + 16 * n
res["traffic_over_algorithmic"] = res["hbm_bytes_per_launch"] / res["algorithmic_bytes"]
json.dump(res, open(out, "w"), indent=1)
if len(sys.argv) > 6:                                   # profiles/spmv_traffic.json: {grid: {"default" | "plain": entry}}
    path, form = sys.argv[5], sys.argv[6]
    try:
        allv = json.load(open(path))
    except Exception:
        allv = {}
    if not isinstance(allv.get(str(grid)), dict) or "hbm_bytes_per_launch" in allv.get(str(grid), {}):
        allv[str(grid)] = {}
    allv[str(grid)][form] = res
    json.dump(allv, open(path, "w"), indent=1)
print(json.dumps(res, indent=1))
