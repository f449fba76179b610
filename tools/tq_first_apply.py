#!/usr/bin/env python3
"""First apply of a fresh ILU(0) preconditioner, timed alone (for timing-only ablation builds whose later applies see stale edge
cells).  usage: KRYST_ILU_GRAPH=0 [KRYST_HIP_LIB=...] tq_first_apply.py [grid=256] [trials=6]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kryst_amd as K
grid = int(sys.argv[1]) if len(sys.argv) > 1 else 256
trials = int(sys.argv[2]) if len(sys.argv) > 2 else 6
ctx = K.Context(0)
a = K.CsrMatrix.stencil7(grid, "aniso", ctx=ctx)
n = a.nrows()
r = ctx.vec(n).fill_splitmix(3); z = ctx.vec(n)
out = []
for _ in range(trials):
    pc = K.TrueIlu0().setup(a)
    ctx.synchronize()
    ctx.timer_start()
    K.check(K.lib().kryst_pc_apply(pc.h, r.h, z.h))
    out.append(ctx.timer_stop())
    del pc
print(f"grid {grid}: first apply of a fresh preconditioner, ms: " + " ".join(f"{v:.4f}" for v in out))
