#!/usr/bin/env python3
"""Single-GPU rehearsal of the multi-rank code path (run with KRYST_FORCE_COMM=1): torch.distributed (gloo) is
imported FIRST exactly as bench.py does for N > 1, a one-rank RCCL communicator is created from a broadcast
unique id, the operator is built through kryst_csr_create_dist, and CG / PCG / BiCGStab run with every inner
product going through the RCCL all-gather + rank fold and every SpMV through the halo-exchange launch sequence.
The results must equal the plain single-GPU path bit for bit."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29577")
import numpy as np
import torch.distributed as dist

dist.init_process_group("gloo", rank=0, world_size=1)
import kryst_amd as K

box = [K.Context.unique_id()]
dist.broadcast_object_list(box, src=0)
ctx_d = K.Context(0, 0, 1, box[0])
ctx_s = K.Context(0)
N = 40
out = {}
for name, ctx in (("dist", ctx_d), ("dist_ipc", ctx_d), ("single", ctx_s)):
    if name == "dist_ipc":
        # the mailbox path of the scalar all-reduce with the REAL RCCL underneath (the handle exchange is an RCCL all-gather; with
        # one rank the kernel sends to and polls its own mailbox): every inner product in one launch, the same bits
        assert ctx.scalar_reduce("ipc") == "ipc", "hipIpc mailbox set-up failed"
    a = K.CsrMatrix.stencil7(N, "poisson", ctx=ctx)
    n = a.nrows()
    b = a.spmv(ctx.vec(n).fill(1.0))
    res = []
    for cls, pc in ((K.CgSolver, None), (K.PcgSolver, K.Jacobi().setup(a)), (K.BiCgStabSolver, None)):
        s = cls(1e-9 if cls is not K.BiCgStabSolver else 1e-6, 300)
        x = ctx.vec(n)
        st = s.solve(a, pc, b, x)
        res.append((st.iterations, st.final_residual, st.converged, tuple(s.residual_history), x.to_host().tobytes()))
    out[name] = res
    ctx.barrier()
    assert ctx.all_reduce(2.5) == 2.5
assert out["dist"] == out["single"], "forced-collective path differs from the plain path"
assert out["dist_ipc"] == out["single"], "mailbox path of the scalar all-reduce differs from the plain path"
print("dist rehearsal ok:", [(r[0], r[2]) for r in out["dist"]])
dist.destroy_process_group()
