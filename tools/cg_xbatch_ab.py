"""CG / PCG iterations per second with x updated in BATCHES (KRYST_CG_X_BATCH = m: x += alpha_i p_i for m iterations in one pass, XBatchOp) against
the fused form that updates x every iteration (m = 1) and the unfused form, interleaved in ONE process on one operator instance.
usage: [XB_LIST=1,2,4,8] cg_xbatch_ab.py [grid=512] [steps=64] [rounds=2]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kryst_amd as K
grid = int(sys.argv[1]) if len(sys.argv) > 1 else 512
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 64
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 2
MS = [int(v) for v in os.environ.get("XB_LIST", "1,2,4,8").split(",")]
ctx = K.Context(0)
a = K.CsrMatrix.stencil7(grid, "poisson", ctx=ctx)
n = a.nrows()
b = a.spmv(ctx.vec(n).fill(1.0))
pcj = K.Jacobi().setup(a)
forms = [("unfused", {"KRYST_CG_FUSE_P": "0", "KRYST_CG_X_BATCH": "1"})] + \
        [(f"fused x-batch {m}", {"KRYST_CG_FUSE_P": "1", "KRYST_SPMV_FUSE_T": "4", "KRYST_CG_X_BATCH": str(m)}) for m in MS]
res = {}
for method, pc in (("cg", None), ("pcg", pcj)):
    for rnd in range(rounds):
        for name, env in forms:
            for k, v in env.items():
                os.environ[k] = v
            x = ctx.vec(n)
            with K.Session(method, a, pc, b, x, tol=0.0, max_iters=16 + steps) as s:
                s.step(16); ctx.synchronize()
                t0 = time.perf_counter(); s.step(steps); ctx.synchronize(); dt = time.perf_counter() - t0
                st = s.end()
            res.setdefault((method, name), []).append((steps / dt, st.final_residual, float(x.to_host()[n // 3])))
    for name, _ in forms:
        v = res[(method, name)]
        print(json.dumps({"grid": grid, "solver": method, "form": name, "iterations_per_s": [round(x[0], 1) for x in v], "best": round(max(x[0] for x in v), 1),
                          "same_residual_and_x_as_unfused": v[0][1:] == res[(method, "unfused")][0][1:]}), flush=True)
