import os, sys, time
sys.path.insert(0, "/root/repo")
import kryst_amd as K
ctx = K.Context(0)
a = K.CsrMatrix.stencil7(128, "poisson", ctx=ctx)
for r in range(2):
    ctx.synchronize(); t0 = time.perf_counter()
    pc = K.Ilup(1).setup(a)
    ctx.synchronize(); print(f"Ilup(1) 128^3 setup #{r}: {1e3 * (time.perf_counter() - t0):.0f} ms", flush=True)
    del pc
for r in range(2):
    ctx.synchronize(); t0 = time.perf_counter()
    pc = K.Ilut(4, 1e-3).setup(a)
    ctx.synchronize(); print(f"Ilut(4,1e-3) 128^3 setup #{r}: {1e3 * (time.perf_counter() - t0):.0f} ms", flush=True)
    del pc
