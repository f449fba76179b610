#!/usr/bin/env python3
"""True ILU(0) set-up of a 7-point grid operator, several times in one process (first call against steady state), with the library's own
phase report (KRYST_ILU_VERBOSE=1).   usage: ilu_setup_times.py [grid=512] [kind=varcoef] [repeats=3]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kryst_amd as K
grid = int(sys.argv[1]) if len(sys.argv) > 1 else 512
kind = sys.argv[2] if len(sys.argv) > 2 else "varcoef"
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
ctx = K.Context(0)
a = K.CsrMatrix.stencil7(grid, kind, ctx=ctx)
for r in range(reps):
    ctx.synchronize(); t0 = time.perf_counter()
    pc = K.TrueIlu0().setup(a)
    ctx.synchronize(); t1 = time.perf_counter()
    del pc
    ctx.synchronize(); t2 = time.perf_counter()
    print(f"{kind} {grid}^3 setup #{r}: {1e3 * (t1 - t0):.1f} ms, destroy {1e3 * (t2 - t1):.1f} ms", flush=True)
