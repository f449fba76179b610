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
configs = []
for variant in (5, 9, 10, 11, 12, 13, 14, 15):
    for group in (1, 16):
        configs.append((variant, 0, 0, group))
configs += [(10, 8, 0, 1), (10, 16, 0, 1), (12, 8, 0, 1)]
res = {c: [] for c in configs}
for r in range(rounds):
    for c in configs:
        os.environ["KRYST_SPMV_VARIANT"] = str(c[0])
        os.environ["KRYST_SPMV_BLOCKS_PER_CU"] = str(c[1])
        os.environ["KRYST_SPMV_SWIZZLE"] = str(c[2])
        os.environ["KRYST_SPMV_GROUP"] = str(c[3])
        res[c].append(a.bench_spmv(x, y, fused_dots=1, reps=20))
print("variant bpc swz grp  median_ms  min_ms   GB/s(median)  frac_of_8TB/s")
for c in sorted(configs, key=lambda c: statistics.median(res[c])):
    med, mn = statistics.median(res[c]), min(res[c])
    print(f"{c[0]:7d} {c[1]:3d} {c[2]:3d} {c[3]:3d}   {med:8.4f} {mn:8.4f}   {b / med / 1e6:9.1f}   {b / med / 1e6 / 8000:.3f}")
