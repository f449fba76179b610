"""The UNFUSED CG / PCG iteration with the direction vectors in a ring and x updated in batches of m iterations (CgDirectionRingOp + XBatchOp:
34 instead of 40 bytes per row and iteration) against the in-place direction pass (m = 1), interleaved in ONE process on one operator instance.
usage: [XB_LIST=1,4,8] [PLAIN=1] cg_ring_ab.py [grid=256] [steps=96] [rounds=2]       (PLAIN=1: the plain 12-bytes-per-entry CSR arrays)"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kryst_amd as K
grid = int(sys.argv[1]) if len(sys.argv) > 1 else 256
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 96
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 2
MS = [int(v) for v in os.environ.get("XB_LIST", "1,4,8").split(",")]
plain = os.environ.get("PLAIN", "0") == "1"
os.environ["KRYST_CG_FUSE_P"] = "0"
if plain:
    os.environ["KRYST_SPMV_COMPRESS"] = "0"
ctx = K.Context(0)
a = K.CsrMatrix.stencil7(grid, "poisson", ctx=ctx)
n = a.nrows()
b = a.spmv(ctx.vec(n).fill(1.0))
pcj = K.Jacobi().setup(a)
res = {}
for method, pc in (("cg", None), ("pcg", pcj)):
    for rnd in range(rounds):
        for m in MS:
            os.environ["KRYST_CG_X_BATCH"] = str(m)
            x = ctx.vec(n)
            with K.Session(method, a, pc, b, x, tol=0.0, max_iters=16 + steps) as s:
                s.step(16); ctx.synchronize()
                t0 = time.perf_counter(); s.step(steps); ctx.synchronize(); dt = time.perf_counter() - t0
                st = s.end()
            res.setdefault((method, m), []).append((steps / dt, st.final_residual, float(x.to_host()[n // 3])))
    for m in MS:
        v = res[(method, m)]
        print(json.dumps({"grid": grid, "operator": "plain CSR" if plain else "default form", "solver": method, "x_batch": m, "iterations_per_s": [round(x[0], 1) for x in v],
                          "best": round(max(x[0] for x in v), 1), "same_residual_and_x_as_m1": v[0][1:] == res[(method, MS[0])][0][1:]}), flush=True)
