#!/usr/bin/env python3
"""Jacobi-PCG on the 64^3 Poisson system (BASELINE configs[0] size) under the SpMV encodings.  usage: small_ab.py [grid=64]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kryst_amd as K
N = int(sys.argv[1]) if len(sys.argv) > 1 else 64
ctx = K.Context(0)
a = K.CsrMatrix.stencil7(N, "poisson", ctx=ctx); n = a.nrows()
b = a.spmv(ctx.vec(n).fill(1.0)); pc = K.Jacobi().setup(a)
for comp in ("0", "1", "2", "3", "3"):
    os.environ["KRYST_SPMV_COMPRESS"] = comp
    best = 0
    for rep in range(5):
        s = K.PcgSolver(1e-8, 1000); x = ctx.vec(n)
        ctx.synchronize(); t0 = time.perf_counter(); st = s.solve(a, pc, b, x); ctx.synchronize(); dt = time.perf_counter() - t0
        best = max(best, st.iterations / dt)
    print(f"grid {N} COMPRESS={comp}: {st.iterations} iterations, best {best:.0f} it/s", flush=True)
