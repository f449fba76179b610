"""spmv_pattern_stage_kernel: runs per XCD group (KRYST_SPMV_STAGE_GROUP) in one process, interleaved.  usage: stage_group_ab.py [grid=512] [groups=1,4,8,16]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kryst_amd as K
grid = int(sys.argv[1]) if len(sys.argv) > 1 else 512
groups = (sys.argv[2] if len(sys.argv) > 2 else "1,4,8,16").split(",")
ctx = K.Context(0); a = K.CsrMatrix.stencil7(grid, "poisson", ctx=ctx); n = a.nrows(); x = ctx.vec(n).fill_splitmix(3); y = ctx.vec(n)
res = {}
for rnd in range(3):
    for g in ["default"] + groups:
        if g == "default": os.environ.pop("KRYST_SPMV_STAGE_GROUP", None)
        else: os.environ["KRYST_SPMV_STAGE_GROUP"] = g
        res.setdefault(g, []).append(sorted(a.bench_spmv(x, y, fused_dots=1, reps=20) for _ in range(3))[1])
for g, v in res.items():
    print(json.dumps({"grid": grid, "group": g, "ms": [round(m, 4) for m in v]}), flush=True)
