#!/usr/bin/env python3
"""Generates tests/golden/krylov_golden.npz from the CPU oracle (oracle/kryst_oracle.c).

The reference (Rust + un-vendored crates) cannot be compiled or run in the build container, and it holds no 3-D
fixtures, so these vectors come from the oracle AFTER it has been pinned by the reference's own known-answer tests
(tests/test_oracle_reference_pins.py).  Each case stores inputs (grid, stencil kind, b) and expected outputs
(x, residual history, iteration count, status) for the strict serial-fold dot order AND for the library's tile
order (T=256, V=2, F=1024).  Run:  python tests/golden/make_golden.py
"""
import os
import sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as O   # noqa: E402

T, V, F = 256, 2, 1024

# name, grid, kind, method, pc, kwargs  -- the five BASELINE.json configs at fixture size (+ the ILU variants)
CASES = [
    ("cfg1_pcg_jacobi_poisson", 12, "poisson", "pcg", "jacobi", dict(tol=1e-8, max_iters=1000)),
    ("cfg2_cg_poisson", 16, "poisson", "cg", None, dict(tol=1e-8, max_iters=2000)),
    ("cfg3_gmres30_left_jacobi_convdiff", 10, "convdiff", "gmres", "jacobi", dict(tol=1e-8, max_iters=120, restart=30, side=O.SIDE_LEFT)),
    ("cfg3b_gmres30_nopc_convdiff", 10, "convdiff", "gmres", None, dict(tol=1e-8, max_iters=120, restart=30)),
    ("cfg4_pcg_jacobi_poisson_2parts", 12, "poisson", "pcg", "jacobi", dict(tol=1e-8, max_iters=3000), 2),
    ("cfg5_bicgstab_aniso", 12, "aniso", "bicgstab", None, dict(tol=None, max_iters=500)),
    ("cfg5b_bicgstab_rpc_ilu0true_aniso", 12, "aniso", "bicgstab_rpc", "ilu0_true", dict(tol=None, max_iters=500)),
    ("cfg5c_bicgstab_rpc_ilup0_aniso", 12, "aniso", "bicgstab_rpc", "ilup0", dict(tol=None, max_iters=500)),
    ("pcg_ilu0compat_poisson", 8, "poisson", "pcg", "ilu0_compat", dict(tol=1e-8, max_iters=300)),
]


def make_pc(name, a):
    return {None: lambda: None, "jacobi": lambda: O.Pc.jacobi(a), "ilu0_true": lambda: O.Pc.ilu0_true(a),
            "ilup0": lambda: O.Pc.ilup0(a), "ilu0_compat": lambda: O.Pc.ilu0_compat(a)}[name]()


def run_case(case, mode):
    name, N, kind, method, pcname, kw = case[:6]
    nparts = case[6] if len(case) > 6 else 1
    a = O.stencil7(N, kind)
    b = a.spmv(np.ones(a.nrows))
    kw = dict(kw)
    if kw.get("tol") is None:
        kw["tol"] = 1e-8 * float(np.linalg.norm(b))          # absolute tolerance (bicgstab.rs:98)
    if mode == "serial":
        rs = O.Reduce.serial()
    else:
        part = None
        if nparts > 1:
            planes = [N * p // nparts for p in range(nparts + 1)]
            part = np.array(planes, dtype=np.int64) * N * N
        rs = O.Reduce.tiled(T, V, F, part_off=part)
    res = O.solve(method, a, b, pc=make_pc(pcname, a), rs=rs, raise_on_error=False, **kw)
    return a, b, kw, res


def main():
    out = {}
    for case in CASES:
        name = case[0]
        for mode in ("serial", "tiled"):
            a, b, kw, res = run_case(case, mode)
            out[f"{name}/{mode}/x"] = res.x
            out[f"{name}/{mode}/history"] = res.history
            out[f"{name}/{mode}/stats"] = np.array([res.iterations, float(res.converged), res.final_residual, res.code])
        out[f"{name}/b"] = b
        out[f"{name}/tol"] = np.array([kw["tol"]])
        print(name, "serial/tiled iterations:", int(out[f"{name}/serial/stats"][0]), int(out[f"{name}/tiled/stats"][0]))
    # one small matrix stored explicitly (CSR of the 4^3 convection-diffusion stencil) + y = A x for a fixed x
    a = O.stencil7(4, "convdiff")
    x = O.splitmix64_uniform(0xC0FFEE, a.ncols)
    out["spmv4/row_ptr"], out["spmv4/col_idx"], out["spmv4/vals"] = a.row_ptr, a.col_idx, a.vals
    out["spmv4/x"], out["spmv4/y"] = x, a.spmv(x)
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "krylov_golden.npz"), **out)


if __name__ == "__main__":
    main()
