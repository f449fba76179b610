#!/usr/bin/env python3
"""FGMRES(restart) + Jacobi on the convection-diffusion stencil, for profiling.  usage: fgmres_only.py [grid=256] [restart=30] [iters=120]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kryst_amd as K

N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
R = int(sys.argv[2]) if len(sys.argv) > 2 else 30
IT = int(sys.argv[3]) if len(sys.argv) > 3 else 120
ctx = K.Context(0)
a = K.CsrMatrix.stencil7(N, "convdiff", ctx=ctx)
n = a.nrows()
b = a.spmv(ctx.vec(n).fill(1.0))
pc = K.Jacobi().setup(a)
for which in ("fgmres", "fgmres-modified", "gmres"):
    for rep in range(2):
        s = K.GmresSolver(R, 1e-30, IT) if which == "gmres" else K.FgmresSolver(1e-30, IT, R)
        if which == "fgmres-modified":
            s = s.with_orthog(K.Orthog.Modified)
        x = ctx.vec(n)
        ctx.synchronize(); t0 = time.perf_counter()
        st = s.solve(a, pc, b, x)
        ctx.synchronize(); dt = time.perf_counter() - t0
    print(json.dumps({"solver": which, "grid": N, "restart": R, "iterations": st.iterations, "seconds": dt,
                      "iterations_per_sec": st.iterations / dt, "last_residual": s.residual_history[-1]}), flush=True)
