#!/usr/bin/env python3
"""ILU-family preconditioners on the level-ordered path: true ILU(0) on operators that are NOT 7-point boxes: setup time with the device-side IKJ factorisation
and with the host loop (KRYST_ILU_DEVICE_SETUP=0), dependency levels, and the apply time of the sync-free triangular kernels
(HIP events, kryst_bench_pc_apply).  One JSON line per operator.   usage: ilu_general.py [N27=96] [nrand=2000000]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import scipy.sparse as sp
import kryst_amd as K

N27 = int(sys.argv[1]) if len(sys.argv) > 1 else 96
NR = int(sys.argv[2]) if len(sys.argv) > 2 else 2000000
ctx = K.Context(0)
rng = np.random.default_rng(1)


def stencil27(N):
    """27-point operator on an N^3 box (trilinear-FE-like): -1 to every neighbour of the 3 x 3 x 3 cube, diagonal 26 + 1."""
    one = sp.diags([np.ones(N - 1), np.ones(N), np.ones(N - 1)], [-1, 0, 1])
    full = sp.kron(one, sp.kron(one, one)).tocsr()
    m = (sp.identity(N ** 3) * 28.0 - full).tocsr()
    m.sort_indices()
    return m


def banded_random(n, per_row, band):
    rows = np.repeat(np.arange(n), per_row)
    cols = np.clip(rows + rng.integers(-band, band + 1, len(rows)), 0, n - 1)
    m = sp.csr_matrix((rng.uniform(-1.0, 1.0, len(rows)), (rows, cols)), shape=(n, n))
    m.sum_duplicates()
    m = m - sp.diags(m.diagonal()) + sp.diags(np.asarray(abs(m).sum(axis=1)).ravel() + 1.0)
    m = m.tocsr(); m.sort_indices()
    return m


for name, m in ((f"27-point stencil, {N27}^3", stencil27(N27)), (f"random band (9 per row, |i-j| <= 2000), {NR} rows", banded_random(NR, 9, 2000))):
    n = m.shape[0]
    a = K.CsrMatrix.from_csr(n, n, m.indptr, m.indices, m.data, ctx=ctx)
    out = {"operator": name, "rows": n, "nnz": int(m.nnz), "spmv_encoding": a.encoding()[0]}
    for dev in ("1", "0"):
        os.environ["KRYST_ILU_DEVICE_SETUP"] = dev
        ctx.synchronize(); t0 = time.perf_counter()
        pc = K.TrueIlu0().setup(a)
        ctx.synchronize()
        out["setup_ms_device_ikj" if dev == "1" else "setup_ms_host_loop"] = (time.perf_counter() - t0) * 1e3
        if dev == "0":
            r = ctx.vec(n).fill_splitmix(3); z = ctx.vec(n)
            out["apply_ms"] = pc.bench_apply(r, z, 10)
            info = pc.ilu_info()
            out["form"] = info["form"]; out["levels_L_U"] = info["levels"]
            out["us_per_level"] = out["apply_ms"] * 1e3 / max(1, sum(info["levels"]))
        del pc
    print(json.dumps(out), flush=True)

# Ilup(1) (fill-in: 13 entries per row on a 7-point operator) and Ilut(4, 1e-3) (magnitude-ordered rows) on 128^3
os.environ["KRYST_ILU_DEVICE_SETUP"] = "1"
for name, kind, mk in (("Ilup(1), 7-point Poisson 128^3", "poisson", lambda: K.Ilup(1)), ("Ilut(4, 1e-3), 7-point anisotropic 128^3", "aniso", lambda: K.Ilut(4, 1e-3))):
    a = K.CsrMatrix.stencil7(128, kind, ctx=ctx)
    n = a.nrows()
    ctx.synchronize(); t0 = time.perf_counter()
    pc = mk().setup(a)
    ctx.synchronize()
    out = {"operator": name, "rows": n, "setup_ms": (time.perf_counter() - t0) * 1e3}
    r = ctx.vec(n).fill_splitmix(3); z = ctx.vec(n)
    out["apply_ms"] = pc.bench_apply(r, z, 10)
    info = pc.ilu_info()
    out["form"] = info["form"]; out["levels_L_U"] = info["levels"]
    out["us_per_level"] = out["apply_ms"] * 1e3 / max(1, sum(info["levels"]))
    print(json.dumps(out), flush=True)
    del pc
