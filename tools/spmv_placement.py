#!/usr/bin/env python3
"""Does the placement of the arrays matter?  The same operator and vectors are created several times in one process (earlier ones
kept alive or freed, padding allocations in between) and the same kernel is timed on each instance.
usage: spmv_placement.py [grid=512] [kind=poisson] [instances=6]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kryst_amd as K

grid = int(sys.argv[1]) if len(sys.argv) > 1 else 512
kind = sys.argv[2] if len(sys.argv) > 2 else "poisson"
inst = int(sys.argv[3]) if len(sys.argv) > 3 else 6
ctx = K.Context(0)
pads = []
for i in range(inst):
    a = K.CsrMatrix.stencil7(grid, kind, ctx=ctx)
    n = a.nrows()
    x = ctx.vec(n).fill_splitmix(0xC0FFEE)
    y = ctx.vec(n)
    b = 12 * a.nnz + 4 * (n + 1) + 16 * n
    t = [a.bench_spmv(x, y, fused_dots=1, reps=20) for _ in range(3)]
    # the kernel's traffic skeleton on the SAME arrays (round 4): does the mix's own ceiling move with the allocation?
    sk = [a.bench_csr_skeleton(x, y, reps=10) for _ in range(3)] if os.environ.get("KRYST_SPMV_COMPRESS") == "0" else [float("nan")]
    print(f"instance {i}: {min(t):.4f} .. {max(t):.4f} ms  {b / min(t) / 1e6 / 8000:.3f}   skeleton {min(sk):.4f} .. {max(sk):.4f} ms  "
          f"kernel / skeleton {min(t) / min(sk):.3f}", flush=True)
    del a, x, y
    pads.append(ctx.vec(1000003 * (i + 1)))          # shifts what the next instance gets
