"""Fused CG form at grid^3: runs per XCD group (KRYST_SPMV_STAGE_GROUP) x tiles per run.  usage: cg_fuse_group_sweep.py [grid=512] [steps=60]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kryst_amd as K
grid = int(sys.argv[1]) if len(sys.argv) > 1 else 512
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 60
ctx = K.Context(0)
a = K.CsrMatrix.stencil7(grid, "poisson", ctx=ctx)
n = a.nrows()
b = a.spmv(ctx.vec(n).fill(1.0))
def run(env):
    for k, v in env.items():
        os.environ[k] = v
    best = 0.0
    for _ in range(2):
        x = ctx.vec(n)
        with K.Session("cg", a, None, b, x, tol=0.0, max_iters=10 + steps) as s:
            s.step(10); ctx.synchronize()
            t0 = time.perf_counter(); s.step(steps); ctx.synchronize(); dt = time.perf_counter() - t0
            s.end()
        best = max(best, steps / dt)
    for k in env: os.environ.pop(k, None)
    return best
print(json.dumps({"grid": grid, "form": "unfused", "it_s": run({"KRYST_CG_FUSE_P": "0"})}), flush=True)
for T in ("4", "2"):
    for g in ("1", "2", "4", "8", "16", "32"):
        print(json.dumps({"grid": grid, "form": "fused", "T": int(T), "group": int(g), "it_s": run({"KRYST_CG_FUSE_P": "1", "KRYST_SPMV_FUSE_T": T, "KRYST_SPMV_STAGE_GROUP": g})}), flush=True)
