#!/usr/bin/env python3
"""Runs the five BASELINE.json configs on one MI355X and prints one JSON object per config (iterations, wall time of the
device-resident solve, iterations/s, convergence).  Evidence for DESIGN.md / profiles; bench.py stays the headline.
usage: bench_configs.py [grid=256] [grid_cfg1=64]   (each config is solved twice; the faster solve is reported, so one-time
work-arena growth is not charged to a solver)"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import kryst_amd as K

grid = int(sys.argv[1]) if len(sys.argv) > 1 else 256
grid1 = int(sys.argv[2]) if len(sys.argv) > 2 else 64
ctx = K.Context(0)


def run(name, N, kind, make_solver, make_pc, abs_tol=False, repeat=2):
    a = K.CsrMatrix.stencil7(N, kind, ctx=ctx)
    n = a.nrows()
    b = a.spmv(ctx.vec(n).fill(1.0))
    bn = K.norm(b)
    t0 = time.perf_counter(); pc = make_pc(a); ctx.synchronize(); t_pc = time.perf_counter() - t0
    best = None
    for _ in range(repeat):
        s = make_solver(1e-8 * bn if abs_tol else 1e-8)
        x = ctx.vec(n)
        ctx.synchronize(); t0 = time.perf_counter()
        st = s.solve(a, pc, b, x)
        ctx.synchronize(); dt = time.perf_counter() - t0
        if best is None or dt < best[0]:
            best = (dt, st, s, x)
    dt, st, s, x = best
    r = a.spmv(x); K.axpy(-1.0, b, r)
    out = {"config": name, "grid": N, "rows": n, "iterations": st.iterations, "converged": st.converged,
           "final_residual": st.final_residual, "true_rel_residual": K.norm(r) / bn, "solve_seconds": dt,
           "iterations_per_sec": st.iterations / dt, "pc_setup_seconds": t_pc}
    print(json.dumps(out), flush=True)


run("1: Jacobi-PCG, Poisson (reference CPU-plumbing size)", grid1, "poisson", lambda t: K.PcgSolver(t, 1000), lambda a: K.Jacobi().setup(a))
run("2: CG, Poisson", grid, "poisson", lambda t: K.CgSolver(t, 2000), lambda a: None)
run("3: GMRES(30) Left+Jacobi (kryst-compat), convection-diffusion", grid, "convdiff",
    lambda t: K.GmresSolver(30, t, 600), lambda a: K.Jacobi().setup(a))
run("3b: GMRES(30) no pc, convection-diffusion", grid, "convdiff", lambda t: K.GmresSolver(30, t, 600), lambda a: None)
run("4(1 GPU): Jacobi-PCG, Poisson", grid, "poisson", lambda t: K.PcgSolver(t, 3000), lambda a: K.Jacobi().setup(a))
run("5 compat: BiCGStab (pc ignored, as the reference), anisotropic Poisson", grid, "aniso", lambda t: K.BiCgStabSolver(t, 3000),
    lambda a: None, abs_tol=True)
run("5 ext: right-preconditioned BiCGStab + true ILU(0), anisotropic Poisson", grid, "aniso",
    lambda t: K.BiCgStabRightPcSolver(t, 3000), lambda a: K.TrueIlu0().setup(a), abs_tol=True)
run("5 ext: right-preconditioned BiCGStab + Ilup(0) as written (SGS), anisotropic Poisson", grid, "aniso",
    lambda t: K.BiCgStabRightPcSolver(t, 3000), lambda a: K.Ilup(0).setup(a), abs_tol=True)
run("f-3: FGMRES(30) + Jacobi (classical GS), convection-diffusion", grid, "convdiff",
    lambda t: K.FgmresSolver(t, 600, 30), lambda a: K.Jacobi().setup(a))
run("f-3: FGMRES(30) + Jacobi (modified), convection-diffusion", grid, "convdiff",
    lambda t: K.FgmresSolver(t, 600, 30).with_orthog(K.Orthog.Modified), lambda a: K.Jacobi().setup(a))
run("f-3: CGS (pc ignored, as the reference), convection-diffusion", grid, "convdiff", lambda t: K.CgsSolver(t, 600), lambda a: None)
run("f-3: TFQMR as written (does not converge; 100 iterations timed), convection-diffusion", grid, "convdiff",
    lambda t: K.TfqmrSolver(1e-30, 100), lambda a: None)
