#!/usr/bin/env python3
"""Streaming-rate probe for the BLAS-1 kernels beyond the 256 MiB Infinity Cache: grid size A/B, interleaved rounds."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kryst_amd as K
ctx = K.Context(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 27
x, y, z = ctx.vec(n), ctx.vec(n), ctx.vec(n)
x.fill_splitmix(1); y.fill_splitmix(2)
def timeit(fn, nbytes, reps=10):
    fn(); ctx.synchronize()
    ctx.timer_start()
    for _ in range(reps): fn()
    return nbytes / (ctx.timer_stop() / reps) / 1e6
ops = {"fill(1W)": (lambda: z.fill(1.0), 8 * n), "axpy(2R1W)": (lambda: K.axpy(0.5, x, y), 24 * n),
       "sub(2R1W)": (lambda: K.lib().kryst_sub(x.h, y.h, z.h), 24 * n), "dot(2R)": (lambda: K.dot(x, y), 16 * n)}
bpcs = [1, 2, 3, 4, 6, 8, 12, 16]
res = {(b, k): [] for b in bpcs for k in ops}
for r in range(5):
    for b in bpcs:
        os.environ["KRYST_EW_BLOCKS_PER_CU"] = str(b)
        for k, (fn, nb) in ops.items():
            res[(b, k)].append(timeit(fn, nb))
print(f"n={n} ({8*n/2**20:.0f} MiB/vec)  median GB/s per blocks/CU")
print("bpc   " + "  ".join(f"{k:>11s}" for k in ops))
for b in bpcs:
    print(f"{b:3d}   " + "  ".join(f"{statistics.median(res[(b, k)]):11.1f}" for k in ops))
print("copy(hipMemcpy D2D)", timeit(lambda: z.copy_from(x), 16 * n))
