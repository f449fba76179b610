#!/usr/bin/env python3
"""ILU apply alone (for rocprofv3 runs and A/B of the triangular-solve forms).  usage: ilu_only.py [grid] [reps] [mode: true|compat]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kryst_amd as K
grid = int(sys.argv[1]) if len(sys.argv) > 1 else 256
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
mode = sys.argv[3] if len(sys.argv) > 3 else "true"
ctx = K.Context(0)
a = K.CsrMatrix.stencil7(grid, "aniso", ctx=ctx)
n = a.nrows()
t0 = time.perf_counter()
pc = (K.TrueIlu0() if mode == "true" else K.Ilu0()).setup(a)
ctx.synchronize()
t_setup = time.perf_counter() - t0
r = ctx.vec(n).fill_splitmix(3)
z = ctx.vec(n)
pc.apply(r, z); pc.apply(r, z); ctx.synchronize()
ctx.timer_start()
for _ in range(reps):
    K.check(K.lib().kryst_pc_apply(pc.h, r.h, z.h))
ms = ctx.timer_stop() / reps
print(f"grid {grid} ILU({mode}) WAVE={os.environ.get('KRYST_ILU_WAVE', 'default')}: setup {t_setup:.2f} s, apply {ms:.4f} ms "
      f"({(12 * a.nnz + 4 * (n + 1) + 24 * n) / ms / 1e6:.0f} GB/s on B_spmv + 8n)")
