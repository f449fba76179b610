#!/usr/bin/env python3
"""Streaming-rate probe for the BLAS-1 kernels at sizes beyond the 256 MiB Infinity Cache."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kryst_amd as K
ctx = K.Context(0)
for n in (1 << 24, 1 << 27, (1 << 27) + 13 * 512 + 2048, 3 * (1 << 25)):
    x, y, z = ctx.vec(n), ctx.vec(n), ctx.vec(n)
    x.fill_splitmix(1); y.fill_splitmix(2)
    def timeit(fn, nbytes, reps=20):
        fn(); ctx.synchronize()
        ctx.timer_start()
        for _ in range(reps): fn()
        ms = ctx.timer_stop() / reps
        return nbytes / ms / 1e6
    r = {
        "fill(1W)": timeit(lambda: z.fill(1.0), 8 * n),
        "copy(1R1W)": timeit(lambda: z.copy_from(x), 16 * n),
        "axpy(2R1W)": timeit(lambda: K.axpy(0.5, x, y), 24 * n),
        "sub(2R1W)": timeit(lambda: K.lib().kryst_sub(x.h, y.h, z.h), 24 * n),
    }
    ms = None
    print(f"n={n} ({8*n/2**20:.0f} MiB/vec): " + "  ".join(f"{k} {v:7.1f} GB/s" for k, v in r.items()))
    del x, y, z
