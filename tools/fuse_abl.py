"""Timing-only ablations of spmv_pattern_fuse_kernel in ONE process (a -DKR_TUNING build: make VARIANT=tuning EXTRA=-DKR_TUNING; KRYST_HIP_LIB selects it).
KRYST_FUSE_ABL bits: 1 no far operands, 2 the window's halo not loaded, 4 no y store, 8 no p_new / x stores.  kryst_bench_spmv_fused: HIP events around
back-to-back launches on vectors of its own (the results are wrong, the timing is what is read).
usage: KRYST_HIP_LIB=.../libkryst_hip_tuning.so fuse_abl.py [grid=512]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kryst_amd as K
grid = int(sys.argv[1]) if len(sys.argv) > 1 else 512
ctx = K.Context(0); a = K.CsrMatrix.stencil7(grid, "poisson", ctx=ctx); n = a.nrows(); x = ctx.vec(n).fill_splitmix(3); y = ctx.vec(n)
for T in ("4", "2"):
    os.environ["KRYST_SPMV_FUSE_T"] = T
    for rnd in range(2):
        for abl in (0, 1, 2, 3, 4, 8, 12, 15):
            os.environ["KRYST_FUSE_ABL"] = str(abl)
            ms = sorted(a.bench_spmv_fused(x, y, reps=10) for _ in range(3))[1]
            print(f"T {T} round {rnd} abl {abl:2d}: {ms:.4f} ms  ({50 * n / ms / 8e9:.3f} of 8 TB/s on the full kernel's 50 B per row)", flush=True)
