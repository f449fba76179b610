#!/bin/bash
# A/B of two builds of the library on CG at grid^3 (fused form forced on), alternating processes on one box.  usage: cg_lib_ab.sh libA libB [grid=512] [rounds=4]
A=$1; B=$2; G=${3:-512}; R=${4:-4}
for r in $(seq $R); do
  for L in $A $B; do
    echo -n "$(basename $L) "
    KRYST_HIP_LIB=$L KRYST_CG_FUSE_P=1 python3 - $G <<'PY'
import os, sys, time
sys.path.insert(0, "/root/repo")
import kryst_amd as K
g = int(sys.argv[1]); ctx = K.Context(0)
a = K.CsrMatrix.stencil7(g, "poisson", ctx=ctx); n = a.nrows(); b = a.spmv(ctx.vec(n).fill(1.0))
best = 0.0
for _ in range(3):
    x = ctx.vec(n)
    with K.Session("cg", a, None, b, x, tol=0.0, max_iters=70) as s:
        s.step(10); ctx.synchronize(); t0 = time.perf_counter(); s.step(60); ctx.synchronize(); dt = time.perf_counter() - t0; s.end()
    best = max(best, 60 / dt)
print(f"{g}^3 CG {best:.1f} it/s")
PY
  done
done
