#!/bin/bash
# timing-only ablations of spmv_pattern_fuse_kernel (builds: make VARIANT=ablN EXTRA=-DKR_FUSE_ABL=N; bits: 1 no far operands, 2 no window halo, 4 no y store,
# 8 no p_new / x stores): the fused kernel's time from the phase run of a CG session, alternating processes.
for r in 1 2; do
for L in "" _abl1 _abl2 _abl3 _abl4 _abl8 _abl15; do
  echo -n "lib$L: "
  KRYST_HIP_LIB=/root/repo/kryst_amd/lib/libkryst_hip$L.so KRYST_CG_FUSE_P=1 python3 - <<'PY'
import sys; sys.path.insert(0, "/root/repo")
import kryst_amd as K
ctx = K.Context(0); a = K.CsrMatrix.stencil7(512, "poisson", ctx=ctx); n = a.nrows(); b = a.spmv(ctx.vec(n).fill(1.0)); x = ctx.vec(n)
with K.Session("cg", a, None, b, x, tol=0.0, max_iters=45) as s:
    s.step(5); ctx.synchronize(); ctx.phase_timing_begin(); s.step(40); ph = ctx.phase_timing_end(); s.end()
print("fused spmv %.4f ms, residual pass %.4f ms" % (ph["spmv"] / 40, ph["blas1_residual"] / 40))
PY
done
done
