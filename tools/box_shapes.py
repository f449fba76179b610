#!/usr/bin/env python3
"""Apply time of the box-stencil wavefront solve (tri_box.h) on 27-point boxes of different shapes: one block of lines with long lines
isolates the cost of a step, many blocks of short lines the cost of a block hop.   usage: box_shapes.py 4096x7x8,96x96x96,..."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import scipy.sparse as sp
import kryst_amd as K

shapes = [tuple(int(v) for v in s.split("x")) for s in (sys.argv[1] if len(sys.argv) > 1 else "4096x7x8,96x96x96").split(",")]
ctx = K.Context(0)


P19 = len(sys.argv) > 2 and sys.argv[2] == "19"      # second argument "19": the 19-point stencil (no corner couplings)


def stencil27(Ni, Nj, Nk):
    if P19:
        n = Ni * Nj * Nk
        idx = np.arange(n); i, j, k = idx % Ni, (idx // Ni) % Nj, idx // (Ni * Nj)
        rows, cols = [idx], [idx]
        for dk in (-1, 0, 1):
            for dj in (-1, 0, 1):
                for di in (-1, 0, 1):
                    if (dk, dj, di) == (0, 0, 0) or abs(dk) + abs(dj) + abs(di) > 2:
                        continue
                    ok = (i + di >= 0) & (i + di < Ni) & (j + dj >= 0) & (j + dj < Nj) & (k + dk >= 0) & (k + dk < Nk)
                    rows.append(idx[ok]); cols.append(idx[ok] + di + Ni * dj + Ni * Nj * dk)
        rows, cols = np.concatenate(rows), np.concatenate(cols)
        m = sp.csr_matrix((np.where(rows == cols, 20.0, -1.0), (rows, cols)), shape=(n, n))
        m.sort_indices()
        return m
    t = lambda N: sp.diags([np.ones(N - 1), np.ones(N), np.ones(N - 1)], [-1, 0, 1]) if N > 1 else sp.identity(1)
    full = sp.kron(t(Nk), sp.kron(t(Nj), t(Ni))).tocsr()
    m = (sp.identity(Ni * Nj * Nk) * 28.0 - full).tocsr()
    m.sort_indices()
    return m


for Ni, Nj, Nk in shapes:
    m = stencil27(Ni, Nj, Nk)
    n = m.shape[0]
    a = K.CsrMatrix.from_csr(n, n, m.indptr, m.indices, m.data, ctx=ctx)
    pc = K.TrueIlu0().setup(a)
    r = ctx.vec(n).fill_splitmix(3); z = ctx.vec(n)
    ms = min(pc.bench_apply(r, z, 10) for _ in range(3))
    info = pc.ilu_info()
    steps = Ni + 29
    print(json.dumps({"box": [Ni, Nj, Nk], "rows": n, "form": info["form"], "apply_ms": ms, "us_per_step_if_one_block": ms * 1e3 / (2 * steps)}), flush=True)
    del pc, a
