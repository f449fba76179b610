#!/usr/bin/env python3
"""Times apply_chebyshev (as a preconditioner object, degree m) alone.  usage: cheb_only.py [grid=256] [m=4] [reps=20]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kryst_amd as K
grid = int(sys.argv[1]) if len(sys.argv) > 1 else 256
m = int(sys.argv[2]) if len(sys.argv) > 2 else 4
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
ctx = K.Context(0)
a = K.CsrMatrix.stencil7(grid, "poisson", ctx=ctx)
n = a.nrows()
pc = K.ChebyshevPc(m, 0.1, 12.0).setup(a)
r = ctx.vec(n).fill_splitmix(3); z = ctx.vec(n)
pc.apply(r, z); ctx.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    pc.apply(r, z)
ctx.synchronize()
dt = (time.perf_counter() - t0) / reps
print(f"grid {grid}: Chebyshev(m={m}) apply {dt * 1e3:.3f} ms = {dt * 1e6 / m:.1f} us per degree")
