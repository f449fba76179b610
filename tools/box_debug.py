#!/usr/bin/env python3
"""Where does the box-stencil solve differ from the oracle?  A 27-point operator with random coefficients, true ILU(0); NaNs are put into
every CU's LDS first (Context.poison_lds), then fresh preconditioners are set up and applied `reps` times; a mismatch is applied again with the
hyperplane kernels to tell a set-up error from an apply error.   usage: box_debug.py NixNjxNk [reps=40]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, scipy.sparse as sp
import kryst_amd as K
from oracle import oracle as O
Ni, Nj, Nk = (int(v) for v in sys.argv[1].split("x"))
rng = np.random.default_rng(80)
n = Ni * Nj * Nk
idx = np.arange(n)
i, j, k = idx % Ni, (idx // Ni) % Nj, idx // (Ni * Nj)
rows, cols, vals = [], [], []
for dk in (-1, 0, 1):
    for dj in (-1, 0, 1):
        for di in (-1, 0, 1):
            if (dk, dj, di) == (0, 0, 0):
                continue
            ok = (i + di >= 0) & (i + di < Ni) & (j + dj >= 0) & (j + dj < Nj) & (k + dk >= 0) & (k + dk < Nk)
            r = idx[ok]
            rows.append(r); cols.append(r + di + Ni * dj + Ni * Nj * dk); vals.append(-rng.uniform(0.2, 1.0, len(r)))
rows, cols, vals = np.concatenate(rows), np.concatenate(cols), np.concatenate(vals)
dsum = np.ones(n)
np.add.at(dsum, rows, np.abs(vals))
m = sp.coo_matrix((np.concatenate([vals, dsum]), (np.concatenate([rows, idx]), np.concatenate([cols, idx]))), shape=(n, n)).tocsr()
m.sort_indices()
ao = O.Csr(n, n, m.indptr, m.indices, m.data)
ctx = K.Context(0)
ctx.poison_lds()
a = K.CsrMatrix.from_csr(n, n, m.indptr, m.indices, m.data, ctx=ctx)
ref = O.Pc.ilu0_true(ao)
r = np.random.default_rng(1).standard_normal(n)
want = ref.apply(r)
# soak: fresh preconditioner + first apply, many times; on a mismatch the same preconditioner is applied again with the hyperplane kernels
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
nbad = 0
for rep in range(reps):
    pc = K.TrueIlu0().setup(a)
    r = np.random.default_rng(rep).standard_normal(n)
    want = ref.apply(r)
    got = pc.apply(r)
    bad = np.flatnonzero(got != want)
    if len(bad):
        nbad += 1
        form = pc.ilu_info()["form"]
        got2 = pc.apply(r)
        os.environ["KRYST_ILU_PLANES"] = "1"
        got3 = pc.apply(r)
        del os.environ["KRYST_ILU_PLANES"]
        b = bad[:6]
        print(f"rep {rep}: {len(bad)} rows differ ({form}); the same apply again: {int((got2 != want).sum())} differ; with the hyperplane kernels: {int((got3 != want).sum())} differ; "
              f"nan {int(np.isnan(got).sum())}; first (i, j, k): {[(int(x % Ni), int((x // Ni) % Nj), int(x // (Ni * Nj))) for x in b]}; "
              f"k range {int((bad // (Ni * Nj)).min())}..{int((bad // (Ni * Nj)).max())}, j range {int(((bad // Ni) % Nj).min())}..{int(((bad // Ni) % Nj).max())}, "
              f"i range {int((bad % Ni).min())}..{int((bad % Ni).max())}; rel {[float(abs(got[x] - want[x]) / abs(want[x])) for x in b[:3]]}", flush=True)
    del pc
print(f"soak: {nbad} of {reps} first applies differed", flush=True)
