#!/usr/bin/env python3
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, scipy.sparse as sp
import kryst_amd as K
from kryst_amd._ffi import lib
Ni, Nj, Nk = (int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "4096x7x8").split("x"))
ctx = K.Context(0)
t = lambda N: sp.diags([np.ones(N - 1), np.ones(N), np.ones(N - 1)], [-1, 0, 1]) if N > 1 else sp.identity(1)
m = (sp.identity(Ni * Nj * Nk) * 28.0 - sp.kron(t(Nk), sp.kron(t(Nj), t(Ni)))).tocsr(); m.sort_indices()
n = m.shape[0]
a = K.CsrMatrix.from_csr(n, n, m.indptr, m.indices, m.data, ctx=ctx)
pc = K.TrueIlu0().setup(a)
r = ctx.vec(n).fill_splitmix(3); z = ctx.vec(n)
print("apply ms", pc.bench_apply(r, z, 3))
buf = (C.c_longlong * 512)()
lib().kryst_debug_tb_trace(buf)
for b in range(8):
    v = buf[8 * b: 8 * b + 8]
    if v[0]:
        st = max(1, v[3])
        print(f"block {b}: {v[0] / st:.0f} cycles per step: take {v[1] / st:.0f}, pub wait {v[6] / st:.0f}, exchange {v[2] / st:.0f}, arithmetic {v[4] / st:.0f}, store + rest {v[5] / st:.0f}")
