#!/usr/bin/env python3
"""Repeats the ILU apply on fresh buffers and checks that every result has the same bits (a race in the wavefront solve would
show as run-to-run differences).  usage: ilu_soak.py"""
import os
import sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kryst_amd as K
ctx = K.Context(0)
for grid, mode in ((256, 2), (200, 1), (129, 0)):
    a = K.CsrMatrix.stencil7(grid, "aniso", ctx=ctx)
    n = a.nrows()
    pc = [K.Ilu0, K.Ilup, K.TrueIlu0][mode]().setup(a)
    r = ctx.vec(n).fill_splitmix(3); z = ctx.vec(n)
    pc.apply(r, z); ref = z.to_host()
    bad = 0
    for it in range(40):
        r2 = ctx.vec(n).fill_splitmix(3); z2 = ctx.vec(n)      # fresh buffers: stale contents differ
        pc.apply(r2, z2)
        if not np.array_equal(z2.to_host(), ref): bad += 1
    print(f"grid {grid} mode {mode}: {bad} of 40 repeated applies differ; finite: {np.isfinite(ref).all()}")
