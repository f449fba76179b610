#!/usr/bin/env python3
"""A/B the SpMV kernel variants in ONE process, interleaved rounds (needs a -DKRYST_TUNING build).
usage: tune_spmv.py [grid] [rounds]"""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kryst_amd as K

grid = int(sys.argv[1]) if len(sys.argv) > 1 else 256
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 5
ctx = K.Context(0)
a = K.CsrMatrix.stencil7(grid, "poisson", ctx=ctx)
n = a.nrows()
x = ctx.vec(n).fill_splitmix(0xC0FFEE)
y = ctx.vec(n)
b = 12 * a.nnz + 4 * (n + 1) + 16 * n
configs = [("2", "0", "7", "0", "1"), ("3", "1", "7", "0", "1"), ("3", "1", "4", "0", "1"), ("3", "1", "7", "1", "1"), ("3", "1", "4", "1", "1"),
           ("3", "1", "7", "0", "4"), ("3", "1", "7", "0", "16"), ("3", "1", "4", "0", "16"), ("3", "0", "4", "0", "1")]
res = {c: [] for c in configs}
for r in range(rounds):
    for c in configs:
        os.environ["KRYST_SPMV_KERNEL"] = c[0]
        os.environ["KRYST_SPMV_COMPRESS"] = c[1]
        os.environ["KRYST_SPMV_SLOTS"] = c[2]
        os.environ["KRYST_SPMV_SWIZZLE"] = c[3]
        os.environ["KRYST_SPMV_GROUP"] = c[4]
        res[c].append(a.bench_spmv(x, y, fused_dots=1, reps=20))
print(f"grid {grid}: kernel compress slots swz grp   median_ms  min_ms   GB/s(algorithmic)  frac_of_8TB/s")
for c in sorted(configs, key=lambda c: statistics.median(res[c])):
    med, mn = statistics.median(res[c]), min(res[c])
    print(f"{c[0]:>7s} {c[1]:>8s} {c[2]:>5s} {c[3]:>3s} {c[4]:>3s}   {med:8.4f} {mn:8.4f}   {b / med / 1e6:9.1f}   {b / med / 1e6 / 8000:.3f}")
sys.exit(0)
print("variant bpc swz grp  median_ms  min_ms   GB/s(median)  frac_of_8TB/s")
for c in sorted(configs, key=lambda c: statistics.median(res[c])):
    med, mn = statistics.median(res[c]), min(res[c])
    print(f"{c[0]:7d} {c[1]:3d} {c[2]:3d} {c[3]:3d}   {med:8.4f} {mn:8.4f}   {b / med / 1e6:9.1f}   {b / med / 1e6 / 8000:.3f}")
