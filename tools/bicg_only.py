#!/usr/bin/env python3
"""BiCGStab (no preconditioner) on the anisotropic Poisson stencil, for A/B measurements.  usage: bicg_only.py [grid=256] [iters=300]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kryst_amd as K
N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
IT = int(sys.argv[2]) if len(sys.argv) > 2 else 300
ctx = K.Context(0)
a = K.CsrMatrix.stencil7(N, "aniso", ctx=ctx)
n = a.nrows()
b = a.spmv(ctx.vec(n).fill(1.0))
for rep in range(2):
    s = K.BiCgStabSolver(1e-300, IT)
    x = ctx.vec(n)
    ctx.synchronize(); t0 = time.perf_counter()
    st = s.solve(a, None, b, x)
    ctx.synchronize(); dt = time.perf_counter() - t0
print(json.dumps({"solver": "bicgstab", "grid": N, "iterations": st.iterations, "iterations_per_sec": st.iterations / dt}))
