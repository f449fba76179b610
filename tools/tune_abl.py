#!/usr/bin/env python3
"""Timing-only ablations of the plain-CSR SpMV (spmv_wave_kernel) -- which part of the kernel costs what.
Needs the tuning build:  make -C kryst_amd/csrc VARIANT=tune EXTRA=-DKR_TUNING ; KRYST_HIP_LIB=kryst_amd/lib/libkryst_hip_tune.so
mask bits: 1 no x gathers, 2 no LDS products / row sums, 4 no y store / fused dot, 8 cacheable load of the dot vector,
16 no load of the dot vector, 32 no workgroup barrier in the fused dot, 64 cacheable y store.   usage: tune_abl.py [grid] [rounds] [fused_dots] [mask,mask,...] [tpw,tpw,...]"""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kryst_amd as K

grid = int(sys.argv[1]) if len(sys.argv) > 1 else 512
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
ctx = K.Context(0)
a = K.CsrMatrix.stencil7(grid, "poisson", ctx=ctx)
n = a.nrows()
x = ctx.vec(n).fill_splitmix(0xC0FFEE)
y = ctx.vec(n)
b = 12 * a.nnz + 4 * (n + 1) + 16 * n
os.environ["KRYST_SPMV_COMPRESS"] = "0"; os.environ["KRYST_SPMV_KERNEL"] = "2"
nq = int(sys.argv[3]) if len(sys.argv) > 3 else 1
masks = [int(m) for m in sys.argv[4].split(",")] if len(sys.argv) > 4 else list(range(8))
tpws = sys.argv[5].split(",") if len(sys.argv) > 5 else ["1"]
os.environ["KRYST_SPMV_ALIGN"] = sys.argv[6] if len(sys.argv) > 6 else "0"
configs = [(sl, nt, abl, tp) for sl in ("4", "7") for nt in ("0", "1") for abl in masks for tp in tpws]
res = {c: [] for c in configs}
for r in range(rounds):
    for c in configs:
        os.environ["KRYST_SPMV_SLOTS"], os.environ["KRYST_SPMV_NT"], os.environ["KRYST_SPMV_ABL"], os.environ["KRYST_SPMV_WAVE_TPW"] = c[0], c[1], str(c[2]), c[3]
        res[c].append(a.bench_spmv(x, y, fused_dots=nq, reps=10))
print(f"grid {grid} nq {nq}: slots nt abl(1 nogather 2 noLDS 4 noY 8 dvec cached 16 no dvec 32 no barrier)   median_ms   GB/s(algorithmic)  frac")
for c in configs:
    med = statistics.median(res[c])
    print(f"   {c[0]:>4s} {c[1]:>2s} {c[2]:3d} tpw {c[3]:>2s}   {med:8.4f}   {b / med / 1e6:9.1f}   {b / med / 1e6 / 8000:.3f}", flush=True)
