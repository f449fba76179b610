"""Ilup(1) / Ilut(4, 1e-3) set-up times on 128^3, several set-ups in one process (KRYST_ILU_VERBOSE=1 prints the phases).
usage: ilup_setup_times.py [repeats=2] [pause_s=0]   (a pause lets the janitor thread hand the host blocks back to the pool)"""
import os, sys, time
sys.path.insert(0, "/root/repo")
import kryst_amd as K
R = int(sys.argv[1]) if len(sys.argv) > 1 else 2
PAUSE = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
ctx = K.Context(0)
a = K.CsrMatrix.stencil7(128, "poisson", ctx=ctx)
for r in range(R):
    ctx.synchronize(); t0 = time.perf_counter()
    pc = K.Ilup(1).setup(a)
    ctx.synchronize(); print(f"Ilup(1) 128^3 setup #{r}: {1e3 * (time.perf_counter() - t0):.0f} ms", flush=True)
    del pc
    time.sleep(PAUSE)
for r in range(R):
    ctx.synchronize(); t0 = time.perf_counter()
    pc = K.Ilut(4, 1e-3).setup(a)
    ctx.synchronize(); print(f"Ilut(4,1e-3) 128^3 setup #{r}: {1e3 * (time.perf_counter() - t0):.0f} ms", flush=True)
    del pc
    time.sleep(PAUSE)
