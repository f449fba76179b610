#!/usr/bin/env python3
"""Debug: P rank threads in ONE process, peer-store halo: which entries of A*1 differ from the RCCL-path result?"""
import os, sys, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import kryst_amd as K
P = int(sys.argv[1]) if len(sys.argv) > 1 else 2
N = int(sys.argv[2]) if len(sys.argv) > 2 else 16
uid = K.Context.unique_id()
bar = threading.Barrier(P)
results = []
def run(rank):
    ctx = K.Context(0, rank, P, uid)
    a = K.CsrMatrix.stencil7(N, "poisson", ctx=ctx)
    n = a.nrows()
    ones = ctx.vec(n).fill(1.0)
    b = a.spmv(ones).to_host()
    bar.wait()
    mode = a.halo_mode("peer")
    ys = [ctx.vec(n) for _ in range(3)]            # KRYST_DEBUG_NOFREE=1: no hipFree (a device-wide synchronisation) inside the loop
    for rep in range(3):
        b2 = (a.spmv(ones, ys[rep]) if os.environ.get("KRYST_DEBUG_NOFREE") == "1" else a.spmv(ones)).to_host()
        bad = np.flatnonzero(b2 != b)
        if len(bad):
            sys.stdout.write(f"rank {rank} mode {mode} rep {rep}: {len(bad)} of {n} differ; first {bad[:6]}; values {b2[bad[:6]]} expected {b[bad[:6]]}; nan {int(np.isnan(b2).sum())}\n")
        results.append(len(bad))
    ctx.barrier()
ts = [threading.Thread(target=run, args=(r,)) for r in range(P)]
[t.start() for t in ts]; [t.join() for t in ts]
print(f"SUMMARY P={P} N={N}: {len(results)} spmv checks, {sum(1 for r in results if r)} with mismatches", flush=True)
