"""Fused CG form (x in batches) at grid^3: tiles per run x runs per XCD group, interleaved rounds in one process.
usage: cg_fuse_knobs.py [grid=512] [steps=96] [rounds=3]   (KNOBS="T:G,T:G,..." default 4:4,4:16,8:4,8:8)"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kryst_amd as K
grid = int(sys.argv[1]) if len(sys.argv) > 1 else 512
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 96
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 3
knobs = [tuple(k.split(":")) for k in os.environ.get("KNOBS", "4:4,4:16,8:4,8:8").split(",")]
ctx = K.Context(0)
a = K.CsrMatrix.stencil7(grid, "poisson", ctx=ctx)
n = a.nrows()
b = a.spmv(ctx.vec(n).fill(1.0))
os.environ["KRYST_CG_FUSE_P"] = "1"
res = {}
for rnd in range(rounds):
    for T, G in knobs:
        os.environ["KRYST_SPMV_FUSE_T"] = T; os.environ["KRYST_SPMV_STAGE_GROUP"] = G
        x = ctx.vec(n)
        with K.Session("cg", a, None, b, x, tol=0.0, max_iters=16 + steps) as s:
            s.step(16); ctx.synchronize()
            t0 = time.perf_counter(); s.step(steps); ctx.synchronize(); dt = time.perf_counter() - t0
            st = s.end()
        res.setdefault((T, G), []).append((steps / dt, st.final_residual))
for (T, G), v in res.items():
    print(json.dumps({"grid": grid, "lib": os.environ.get("KRYST_HIP_LIB", "default"), "T": int(T), "group": int(G), "it_s": [round(x[0], 1) for x in v], "residual": v[0][1]}), flush=True)
