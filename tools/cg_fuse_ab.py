"""CG / PCG iterations per second with the direction pass inside the SpMV (spmv_pattern_fuse_kernel, runs of 2 / 4 tiles) against the unfused form
(KRYST_CG_FUSE_P=0), interleaved in ONE process on one operator instance (the knobs are read per session step).
usage: cg_fuse_ab.py [grid=512] [steps=60] [rounds=3]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kryst_amd as K
grid = int(sys.argv[1]) if len(sys.argv) > 1 else 512
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 60
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 3
ctx = K.Context(0)
a = K.CsrMatrix.stencil7(grid, "poisson", ctx=ctx)
n = a.nrows()
b = a.spmv(ctx.vec(n).fill(1.0))
pcj = K.Jacobi().setup(a)
forms = [("unfused", {"KRYST_CG_FUSE_P": "0"}), ("fused T=2", {"KRYST_CG_FUSE_P": "1", "KRYST_SPMV_FUSE_T": "2"}), ("fused T=4", {"KRYST_CG_FUSE_P": "1", "KRYST_SPMV_FUSE_T": "4"})]
res = {}
for method, pc in (("cg", None), ("pcg", pcj)):
    for rnd in range(rounds):
        for name, env in forms:
            for k, v in env.items():
                os.environ[k] = v
            x = ctx.vec(n)
            with K.Session(method, a, pc, b, x, tol=0.0, max_iters=10 + steps) as s:
                s.step(10); ctx.synchronize()
                t0 = time.perf_counter(); s.step(steps); ctx.synchronize(); dt = time.perf_counter() - t0
                st = s.end()
            res.setdefault((method, name), []).append((steps / dt, st.final_residual))
    for name, _ in forms:
        v = res[(method, name)]
        print(json.dumps({"grid": grid, "solver": method, "form": name, "iterations_per_s": [round(x[0], 1) for x in v], "best": max(x[0] for x in v),
                          "final_residual": v[0][1], "same_residual_as_unfused": v[0][1] == res[(method, "unfused")][0][1]}), flush=True)
