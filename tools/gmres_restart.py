"""BASELINE config 3 (GMRES + Jacobi on the 256^3 convection-diffusion operator, tol 1e-8): which restart length / iteration budget reaches
the tolerance?  GMRES(30) -- the config as written -- stagnates within its 600 iterations in every preconditioning form (bench.py:
config3_gmres30_jacobi_256.forms).  One JSON line per run: restart, form, iterations, converged, true relative residual, seconds.

    python tools/gmres_restart.py [grid=256] [max_iters=3000] [restarts=30,60,100,200]
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kryst_amd as K          # noqa: E402


def main():
    grid = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    max_iters = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
    restarts = [int(r) for r in (sys.argv[3] if len(sys.argv) > 3 else "30,60,100,200").split(",")]
    ctx = K.Context(0)
    a = K.CsrMatrix.stencil7(grid, "convdiff", ctx=ctx)
    n = a.nrows()
    b = a.spmv(ctx.vec(n).fill(1.0))
    bn = K.norm(b)
    pc = K.Jacobi().setup(a)
    for restart in restarts:
        for label, side in (("left_reference", K.Preconditioning.Left), ("left_textbook_extension", K.Preconditioning.LeftTextbook),
                            ("right_reference", K.Preconditioning.Right)):
            if label != "left_textbook_extension" and restart != 30:
                continue                                   # (the reference's forms at the config's own restart only)
            x = ctx.vec(n)
            s = K.GmresSolver(restart, 1e-8, max_iters).with_preconditioning(side)
            ctx.synchronize(); t0 = time.perf_counter()
            st = s.solve(a, pc, b, x)
            ctx.synchronize(); dt = time.perf_counter() - t0
            r = a.spmv(x); K.sub(b, r, r)
            print(json.dumps({"grid": grid, "restart": restart, "form": label, "max_iters": max_iters, "iterations": st.iterations,
                              "converged_flag": bool(st.converged), "final_residual": st.final_residual, "true_relative_residual": K.norm(r) / bn,
                              "seconds": dt, "iterations_per_s": st.iterations / dt}), flush=True)
            del x, r
    # BiCGStab on the same system for scale (the reference's BiCgStabSolver ignores pc; absolute tolerance 1e-8 ||b||)
    x = ctx.vec(n)
    s = K.BiCgStabSolver(1e-8 * bn, max_iters)
    ctx.synchronize(); t0 = time.perf_counter()
    st = s.solve(a, None, b, x)
    ctx.synchronize(); dt = time.perf_counter() - t0
    r = a.spmv(x); K.sub(b, r, r)
    print(json.dumps({"grid": grid, "solver": "bicgstab", "iterations": st.iterations, "converged_flag": bool(st.converged),
                      "true_relative_residual": K.norm(r) / bn, "seconds": dt}), flush=True)


if __name__ == "__main__":
    main()
