#!/usr/bin/env python3
"""A/B of SpMV launch knobs inside ONE process (the knobs are read per launch): the configurations are timed in turn, several rounds,
and the median per configuration is printed -- box-to-box and run-to-run noise (3-5 %) is larger than most of the effects looked for.
usage: spmv_ab.py [grid=512] [kind=poisson] [rounds=7] [reps=20] -- "A=1 B=2" "A=0" ...   (each argument one configuration; "" = defaults)"""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kryst_amd as K

args = sys.argv[1:]
cfgs = args[args.index("--") + 1:] if "--" in args else [""]
pos = args[:args.index("--")] if "--" in args else args
grid = int(pos[0]) if len(pos) > 0 else 512
kind = pos[1] if len(pos) > 1 else "poisson"
rounds = int(pos[2]) if len(pos) > 2 else 7
reps = int(pos[3]) if len(pos) > 3 else 20
ctx = K.Context(0)
a = K.CsrMatrix.stencil7(grid, kind, ctx=ctx)
n = a.nrows()
x = ctx.vec(n).fill_splitmix(0xC0FFEE)
y = ctx.vec(n)
b = 12 * a.nnz + 4 * (n + 1) + 16 * n
names = sorted({kv.split("=")[0] for c in cfgs for kv in c.split()})
times = {c: [] for c in cfgs}
ref = None
for r in range(rounds):
    for c in cfgs:
        for nm in names:
            os.environ.pop(nm, None)
        for kv in c.split():
            k, v = kv.split("=")
            os.environ[k] = v
        times[c].append(a.bench_spmv(x, y, fused_dots=1, reps=reps))
        got = y.to_host()
        if ref is None:
            ref = got
        assert (got == ref).all(), f"configuration {c!r} changes the result"
for c in cfgs:
    t = times[c]
    med = statistics.median(t)
    print(f"grid {grid} {kind} [{c or 'defaults'}]: median {med:.4f} ms (min {min(t):.4f}, max {max(t):.4f})  {b / med / 1e6 / 8000:.3f} of 8 TB/s on the SURVEY bytes", flush=True)
