#!/usr/bin/env python3
"""What the multi-rank code path costs per CG iteration, on ONE GPU (KRYST_FORCE_COMM=1: a one-rank RCCL communicator; the operator is
built through the distributed path, every inner product goes through the all-gather + rank fold or the hipIpc mailboxes, every SpMV
through the halo launch sequence): the same 256^3-per-rank problem a rank of an 8-GPU 512^3 run has.   usage: dist_overhead.py [grid=256]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("KRYST_FORCE_COMM", "1")
import kryst_amd as K

grid = int(sys.argv[1]) if len(sys.argv) > 1 else 256
uid = K.Context.unique_id()
ctx_d = K.Context(0, 0, 1, uid)
ctx_s = K.Context(0)
for name, ctx, mode, halo in (("single-GPU path", ctx_s, None, None), ("collective path, RCCL all-gather, RCCL halo", ctx_d, "rccl", "rccl"),
                             ("collective path, hipIpc mailboxes, RCCL halo", ctx_d, "ipc", "rccl"),
                             ("collective path, hipIpc mailboxes, halo by peer stores", ctx_d, "ipc", "peer")):
    if mode and ctx.scalar_reduce(mode) != mode:
        print(f"{name}: unavailable"); continue
    a = K.CsrMatrix.stencil7(grid, "poisson", ctx=ctx)
    if halo and a.halo_mode(halo) != halo:
        print(f"{name}: unavailable"); continue
    n = a.nrows()
    b = a.spmv(ctx.vec(n).fill(1.0))
    best = None
    for rep in range(3):
        x = ctx.vec(n)
        with K.Session("cg", a, None, b, x, tol=0.0, max_iters=10 ** 6) as s:
            s.step(10); ctx.synchronize()
            t0 = time.perf_counter(); s.step(100); ctx.synchronize()
            dt = (time.perf_counter() - t0) / 100
        best = dt if best is None else min(best, dt)
    print(f"{name}: {best * 1e3:.4f} ms per CG iteration ({1 / best:.0f} it/s)", flush=True)
    del a, b
