#!/usr/bin/env python3
"""A CG stepping session only (for rocprofv3 --pmc / --kernel-trace runs of the iteration's kernels).  usage: cg_only.py [grid=512] [steps=12] [solver=cg]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kryst_amd as K
grid = int(sys.argv[1]) if len(sys.argv) > 1 else 512
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 12
solver = sys.argv[3] if len(sys.argv) > 3 else "cg"
ctx = K.Context(0)
a = K.CsrMatrix.stencil7(grid, "poisson", ctx=ctx)
n = a.nrows()
b = a.spmv(ctx.vec(n).fill(1.0)); x = ctx.vec(n)
pc = K.Jacobi().setup(a) if solver == "pcg" else None
with K.Session(solver, a, pc, b, x, tol=0.0, max_iters=steps) as s:
    s.step(steps); st = s.end()
print(f"{solver} {grid}^3 {st.iterations} iterations, residual {st.final_residual}")
for _ in range(5):
    K.dot(b, x)
print(f"calibration: ew_kernel<DotOp> reads {2 * n * 8} bytes per launch")
