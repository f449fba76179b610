#!/usr/bin/env python3
"""In-process A/B of whole solver iterations: a stepping session (kryst_session_*) runs `iters` iterations per sample, the settings
(environment variables the library reads per launch, e.g. KRYST_BPC_<Op> = workgroups per CU of one elementwise op, KRYST_CG_DEFER_X)
are timed in turn, several rounds, median per setting.
Methods without a stepping form (gmres, gmres_jacobi, fgmres, bicgstab_ilu = right-preconditioned BiCGStab + true ILU(0)) are timed as
whole solves of `iters` iterations on device vectors (tolerance 0).
usage: solver_ab.py <cg|pcg|bicgstab|cgs|tfqmr|gmres|gmres_jacobi|fgmres|bicgstab_ilu> [grid=256] [kind=poisson] [iters=40] [rounds=5] -- "A=1" "A=2 B=3" ...   ("" = defaults)"""
import os, statistics, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import kryst_amd as K

args = sys.argv[1:]
cfgs = args[args.index("--") + 1:] if "--" in args else [""]
pos = args[:args.index("--")] if "--" in args else args
method = pos[0]
grid = int(pos[1]) if len(pos) > 1 else 256
kind = pos[2] if len(pos) > 2 else "poisson"
iters = int(pos[3]) if len(pos) > 3 else 40
rounds = int(pos[4]) if len(pos) > 4 else 5
ctx = K.Context(0)
a = K.CsrMatrix.stencil7(grid, kind, ctx=ctx)
n = a.nrows()
b = a.spmv(ctx.vec(n).fill(1.0))
pc = K.Jacobi().setup(a) if method in ("pcg", "gmres_jacobi", "fgmres") else (K.TrueIlu0().setup(a) if method == "bicgstab_ilu" else None)
WHOLE = {"gmres": lambda: K.GmresSolver(30, 0.0, iters), "gmres_jacobi": lambda: K.GmresSolver(30, 0.0, iters),
         "fgmres": lambda: K.FgmresSolver(0.0, iters, 30), "bicgstab_ilu": lambda: K.BiCgStabRightPcSolver(0.0, iters)}
names = sorted({kv.split("=")[0] for c in cfgs for kv in c.split()})
times = {c: [] for c in cfgs}
for r in range(rounds):
    for c in cfgs:
        for nm in names:
            os.environ.pop(nm, None)
        for kv in c.split():
            k, v = kv.split("=")
            os.environ[k] = v
        x = ctx.vec(n)
        if method in WHOLE:
            sol = WHOLE[method]()
            ctx.synchronize(); t0 = time.perf_counter()
            try:
                st = sol.solve(a, pc, b, x)
                done = st.iterations
            except K.KError as e:
                done = e.stats.iterations if getattr(e, "stats", None) else iters
            ctx.synchronize()
            times[c].append((time.perf_counter() - t0) / max(1, done) * 1e3)
            continue
        with K.Session(method, a, pc, b, x, tol=0.0, max_iters=10 ** 6) as s:
            s.step(5); ctx.synchronize()
            t0 = time.perf_counter()
            s.step(iters); ctx.synchronize()
            times[c].append((time.perf_counter() - t0) / iters * 1e3)
for c in cfgs:
    t = times[c]
    med = statistics.median(t)
    print(f"{method} {grid}^3 {kind} [{c or 'defaults'}]: median {med:.4f} ms/iteration (min {min(t):.4f}, max {max(t):.4f})  {1e3 / med:.1f} it/s", flush=True)
