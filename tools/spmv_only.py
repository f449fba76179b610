#!/usr/bin/env python3
"""Launch only the SpMV kernel (for rocprofv3 --pmc / --kernel-trace runs and variant tuning).
usage: spmv_only.py [grid] [reps] [fused_dots] [kind: poisson|aniso|convdiff|varcoef]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kryst_amd as K

grid = int(sys.argv[1]) if len(sys.argv) > 1 else 256
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
nq = int(sys.argv[3]) if len(sys.argv) > 3 else 1
kind = sys.argv[4] if len(sys.argv) > 4 else "poisson"
ctx = K.Context(0)
a = K.CsrMatrix.stencil7(grid, kind, ctx=ctx)
n = a.nrows()
x = ctx.vec(n).fill_splitmix(0xC0FFEE)
y = ctx.vec(n)
ms = a.bench_spmv(x, y, fused_dots=nq, reps=reps)
b = 12 * a.nnz + 4 * (n + 1) + 16 * n
print(f"grid {grid} {kind} ({a.encoding()[0]}) nq {nq}: {ms:.4f} ms/launch  {b / ms / 1e6:.1f} GB/s  ({b / ms / 1e6 / 8000:.3f} of 8 TB/s)")
# calibration launches with a KNOWN byte count (MI355X_MICROARCH.md, HBM: calibrate FETCH_SIZE on your own access pattern):
# ew_kernel<DotOp> reads 2*n*8 bytes with 16 B/lane loads and writes n/512*8 bytes
for _ in range(5):
    K.dot(x, y)
print(f"calibration: ew_kernel<DotOp> reads {2 * n * 8} bytes per launch")
