#!/usr/bin/env python3
"""True ILU(0) apply of a 27-point operator alone (for rocprofv3 runs of the box-stencil wavefront kernels).  usage: box_only.py [N=96] [reps=20]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import scipy.sparse as sp
import kryst_amd as K
N = int(sys.argv[1]) if len(sys.argv) > 1 else 96
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
ctx = K.Context(0)
one = sp.diags([np.ones(N - 1), np.ones(N), np.ones(N - 1)], [-1, 0, 1])
m = (sp.identity(N ** 3) * 28.0 - sp.kron(one, sp.kron(one, one))).tocsr()
m.sort_indices()
n = m.shape[0]
a = K.CsrMatrix.from_csr(n, n, m.indptr, m.indices, m.data, ctx=ctx)
pc = K.TrueIlu0().setup(a)
r = ctx.vec(n).fill_splitmix(3); z = ctx.vec(n)
pc.apply(r, z); pc.apply(r, z); ctx.synchronize()
ctx.timer_start()
for _ in range(reps):
    K.check(K.lib().kryst_pc_apply(pc.h, r.h, z.h))
ms = ctx.timer_stop() / reps
print(f"27-point {N}^3 true ILU(0): apply {ms:.4f} ms, {pc.ilu_info()}")
