#!/usr/bin/env python3
"""True ILU(0) set-up of a 7-point grid operator, several times in one process -- destroy, set up again, the way a caller re-factors (Ilup::setup is
called per matrix, ilup.rs:77-134) -- with the library's own phase report (KRYST_ILU_VERBOSE=1).  `benchlike` first does what bench.py has done by
the time it reaches this block (a 512^3 Poisson operator, CG iterations, a GMRES(30) solve: tens of GB allocated and freed before the first set-up).
KRYST_DEV_POOL_MB=0 switches the device block pool off (round 4's behaviour: every destroy gives its blocks back to the driver).
usage: ilu_setup_times.py [grid=512] [kind=varcoef] [repeats=3] [benchlike]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kryst_amd as K
grid = int(sys.argv[1]) if len(sys.argv) > 1 else 512
kind = sys.argv[2] if len(sys.argv) > 2 else "varcoef"
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
benchlike = len(sys.argv) > 4 and sys.argv[4] == "benchlike"
ctx = K.Context(0)
print(f"# KRYST_DEV_POOL_MB={os.environ.get('KRYST_DEV_POOL_MB', '(default 65536)')} benchlike={benchlike}", flush=True)
if benchlike:
    p = K.CsrMatrix.stencil7(512, "poisson", ctx=ctx)
    b = p.spmv(ctx.vec(p.nrows()).fill(1.0)); x = ctx.vec(p.nrows())
    with K.Session("cg", p, None, b, x, tol=0.0, max_iters=60) as s:
        s.step(60); s.end()
    K.GmresSolver(30, 0.0, 30).solve(p, K.Jacobi().setup(p), b, x.fill(0.0))
    del p, b, x
a = K.CsrMatrix.stencil7(grid, kind, ctx=ctx)
for r in range(reps):
    ctx.synchronize(); t0 = time.perf_counter()
    pc = K.TrueIlu0().setup(a)
    ctx.synchronize(); t1 = time.perf_counter()
    del pc
    ctx.synchronize(); t2 = time.perf_counter()
    print(f"{kind} {grid}^3 setup #{r}: {1e3 * (t1 - t0):.1f} ms, destroy {1e3 * (t2 - t1):.1f} ms", flush=True)
t0 = time.perf_counter(); freed = ctx.trim(); t1 = time.perf_counter()
print(f"kryst_ctx_trim: {freed / 1e9:.2f} GB back to the driver in {1e3 * (t1 - t0):.1f} ms", flush=True)
