#!/usr/bin/env python3
"""ILU(0) apply on an Ni x Nj x Nk box (7-point Poisson, natural order): separates the per-step cost of the wavefront kernel
(one block: Nj = Nk = 8, long lines) from the block-to-block start-up cost.  usage: ilu_box.py Ni Nj Nk [reps=10]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, scipy.sparse as sp
import kryst_amd as K
Ni, Nj, Nk = (int(v) for v in sys.argv[1:4])
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 10
def lap(n): return sp.diags([-np.ones(n - 1), 2 * np.ones(n), -np.ones(n - 1)], [-1, 0, 1])
A = (sp.kron(sp.eye(Nk), sp.kron(sp.eye(Nj), lap(Ni))) + sp.kron(sp.eye(Nk), sp.kron(lap(Nj), sp.eye(Ni)))
     + sp.kron(lap(Nk), sp.kron(sp.eye(Nj), sp.eye(Ni)))).tocsr()
A.sort_indices()
ctx = K.Context(0)
a = K.CsrMatrix.from_csr(A.shape[0], A.shape[1], A.indptr.astype(np.int64), A.indices.astype(np.int64), A.data, ctx=ctx)
pc = K.TrueIlu0().setup(a)
n = A.shape[0]
r = ctx.vec(n).fill_splitmix(3); z = ctx.vec(n)
pc.apply(r, z); ctx.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    pc.apply(r, z)
ctx.synchronize()
dt = (time.perf_counter() - t0) / reps
print(f"box {Ni}x{Nj}x{Nk}: ILU apply {dt * 1e3:.3f} ms; per factor {dt * 5e2:.3f} ms = {dt * 5e5 / (Ni + 14):.3f} us per step of the longest line")
