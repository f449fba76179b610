#!/usr/bin/env python3
"""Times the ILU(0) triangular-solve apply alone.  usage: ilu_only.py [grid] [mode 0|1|2] [reps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kryst_amd as K
grid = int(sys.argv[1]) if len(sys.argv) > 1 else 256
mode = int(sys.argv[2]) if len(sys.argv) > 2 else 2
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
ctx = K.Context(0)
a = K.CsrMatrix.stencil7(grid, "aniso", ctx=ctx)
n = a.nrows()
t0 = time.perf_counter()
pc = [K.Ilu0, K.Ilup, K.TrueIlu0][mode]().setup(a)
print(f"setup {time.perf_counter() - t0:.2f} s")
r = ctx.vec(n).fill_splitmix(3); z = ctx.vec(n)
pc.apply(r, z); ctx.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    pc.apply(r, z)
t_host = (time.perf_counter() - t0) / reps
ctx.synchronize()
dt = (time.perf_counter() - t0) / reps
print(f"host enqueue time per apply {t_host * 1e3:.3f} ms")
print(f"grid {grid}: ILU apply {dt * 1e3:.3f} ms  ({3 * grid - 2} levels per factor -> {dt * 1e6 / (2 * (3 * grid - 2)):.2f} us per level)")
