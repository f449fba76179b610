#!/usr/bin/env python3
"""Per-block / per-quadrant timeline of the forward 16 x 16 wavefront solve (tri_quad.h).  Needs the trace build:
make -C kryst_amd/csrc VARIANT=trace EXTRA=-DKR_TW_TRACE ; KRYST_HIP_LIB=kryst_amd/lib/libkryst_hip_trace.so.   usage: tq_trace.py N"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import kryst_amd as K
from kryst_amd import _ffi
N = int(sys.argv[1])
ctx = K.Context(0)
a = K.CsrMatrix.stencil7(N, "aniso", ctx=ctx)
pc = K.TrueIlu0().setup(a)
n = a.nrows()
r = ctx.vec(n).fill_splitmix(3); z = ctx.vec(n)
for _ in range(int(os.environ.get('TQ_APPLIES', '3'))):
    K.check(K.lib().kryst_pc_apply(pc.h, r.h, z.h)) if False else None
    import ctypes
    from kryst_amd._ffi import lib as _l
    _l().kryst_pc_apply(pc.h, r.h, z.h); ctx.synchronize()
nbj = (N + 15) // 16
nb = nbj * nbj
nch = (N + 14 + 7) // 8
buf = (C.c_longlong * (nb * 64))()
lib = _ffi.lib()
lib.kryst_debug_tq_trace.argtypes = [C.POINTER(C.c_longlong), C.c_int32]
assert lib.kryst_debug_tq_trace(buf, nb * 64) == 0
t = np.array(buf, dtype=np.int64).reshape(nb, 4, 16).astype(np.float64) / 100.0      # us
if os.environ.get("TQ_DUMP"):
    np.save(os.environ["TQ_DUMP"], t)                      # the whole table [block][quadrant][slot], us
t0 = t[:, :, 0].min()
print(f"N {N}: {nb} blocks, {nch} chunks per block; times in us after the first entry")
print("block  J  K  q | entry  chunk0 chunk1 chunk2 chunk3 ... end | wait: nbr stage ring coef store steps fetch | us/chunk (chunks 4..end)")
show = [b for b in range(nb) if b < 3 or b % nbj == b // nbj or b == nb - 1]
if len(sys.argv) > 2:
    show = [int(v.split(',')[1]) * nbj + int(v.split(',')[0]) for v in sys.argv[2:]]
for b in show:
    for q in range(4):
        x = t[b, q]
        rate = (x[10] - x[5]) / max(nch - 4, 1)
        print(f"{b:5d} {b % nbj:2d} {b // nbj:2d} {q:2d} | {x[0] - t0:6.1f} " + " ".join(f"{x[2 + c] - t0:6.1f}" for c in range(4)) +
              f" ... {x[10] - t0:7.1f} | {x[11]:6.1f} {x[12]:6.1f} {x[13]:6.1f} {x[14]:6.1f} {x[15]:6.1f} {x[1]:6.1f} {x[9]:6.1f} | {rate:5.2f}")
ends = t[:, :, 10] - t0
print(f"last end {ends.max():.1f} us; block (0,0) q0 ends {ends[0, 0]:.1f}; diagonal block ends: " + " ".join(f"{ends[d * nbj + d, 3]:.0f}" for d in range(nbj)))
