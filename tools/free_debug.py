#!/usr/bin/env python3
"""true ILU(0) apply of a 27-point operator on an Ni x Nj x Nk box through the level-ordered forms (KRYST_ILU_BOX=0: rows of 13 entries, chains of
virtual rows in tri_run_free_kernel) against the oracle: where do the results differ?   usage: free_debug.py [Ni Nj Nk]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["KRYST_ILU_BOX"] = "0"
import numpy as np, scipy.sparse as sp
import kryst_amd as K
from oracle import oracle as O
Ni, Nj, Nk = (int(x) for x in sys.argv[1:4]) if len(sys.argv) > 3 else (41, 30, 19)
t = lambda n: sp.diags([np.ones(n - 1), np.ones(n), np.ones(n - 1)], [-1, 0, 1])
rng = np.random.default_rng(3)
m = (sp.identity(Ni * Nj * Nk) * 28.0 - sp.kron(t(Nk), sp.kron(t(Nj), t(Ni)))).tocsr(); m.sort_indices()
m.data = m.data * rng.uniform(0.5, 1.5, len(m.data))
n = m.shape[0]
a_o = O.Csr(n, n, m.indptr, m.indices, m.data)
ctx = K.Context(0)
a = K.CsrMatrix.from_csr(n, n, m.indptr, m.indices, m.data, ctx=ctx)
ref = O.Pc.ilu0_true(a_o)
for knobs in [{}] + [dict(kv.split("=") for kv in arg.split(",")) for arg in sys.argv[4:]]:
    for k, v in knobs.items(): os.environ[k] = v
    pc = K.TrueIlu0().setup(a)
    r = np.random.default_rng(5).standard_normal(n)
    z, zr = pc.apply(r), ref.apply(r)
    bad = np.flatnonzero(~((z == zr) | (np.isnan(z) & np.isnan(zr))))
    print(knobs, pc.ilu_info()["form"], pc.ilu_info()["levels"], "differing:", len(bad), "nan:", int(np.isnan(z).sum()), "first/last differing rows:", bad[:5], bad[-5:], flush=True)
    del pc
    for k in knobs: del os.environ[k]
