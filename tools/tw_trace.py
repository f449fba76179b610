#!/usr/bin/env python3
"""Per-block timeline of the forward wavefront solve (needs the trace build: make VARIANT=trace EXTRA=-DKR_TW_TRACE and
KRYST_HIP_LIB=kryst_amd/lib/libkryst_hip_trace.so).  usage: tw_trace.py Ni Nj Nk"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, scipy.sparse as sp
import kryst_amd as K
from kryst_amd import _ffi
Ni, Nj, Nk = (int(v) for v in sys.argv[1:4])
def lap(n): return sp.diags([-np.ones(n - 1), 2 * np.ones(n), -np.ones(n - 1)], [-1, 0, 1])
A = (sp.kron(sp.eye(Nk), sp.kron(sp.eye(Nj), lap(Ni))) + sp.kron(sp.eye(Nk), sp.kron(lap(Nj), sp.eye(Ni)))
     + sp.kron(lap(Nk), sp.kron(sp.eye(Nj), sp.eye(Ni)))).tocsr()
A.sort_indices()
ctx = K.Context(0)
a = K.CsrMatrix.from_csr(A.shape[0], A.shape[1], A.indptr.astype(np.int64), A.indices.astype(np.int64), A.data, ctx=ctx)
pc = K.TrueIlu0().setup(a)
n = A.shape[0]
r = ctx.vec(n).fill_splitmix(3); z = ctx.vec(n)
for _ in range(3):
    pc.apply(r, z); ctx.synchronize()
nb = ((Nj + 7) // 8) * ((Nk + 7) // 8)
buf = (C.c_longlong * (8 * nb))()
lib = _ffi.lib() if hasattr(_ffi, "lib") else _ffi.LIB
lib.kryst_debug_tw_trace.argtypes = [C.POINTER(C.c_longlong), C.c_int32]
assert lib.kryst_debug_tw_trace(buf, 8 * nb) == 0
t = np.array(buf, dtype=np.int64).reshape(nb, 8)
t0 = t[:, 0].min()
nbj = (Nj + 7) // 8
print("block  J  K   entry gate-open first-pub(m) chunk1-done   end   (us)   poller: rounds empty us/round")
for b in range(nb):
    e, f, end, sp_, po, c8, c16 = t[b, 0], t[b, 1], t[b, 2], t[b, 3], t[b, 4], t[b, 5], t[b, 6]
    if nb <= 64 or b % nbj in (0, 1, nbj - 1) or b // nbj in (0, 1):
        print(f"{b:5d} {b % nbj:2d} {b // nbj:2d} {(e - t0) / 100:7.1f} {(c8 - t0) / 100:7.1f} {(c16 - t0) / 100:7.1f}({t[b, 7]:2d}) {(f - t0) / 100:7.1f} {(end - t0) / 100:7.1f}        {sp_ // 1000000:6d} {sp_ % 1000000:6d} {po / 100 / max(sp_ // 1000000, 1):8.2f}")

late = [(b % nbj, b // nbj) for b in range(1, nb) if t[b, 5] - t[b, 0] < 100]
print(f"blocks whose gate was open within 1 us of their entry (they were waiting for a slot, not for data): {len(late)} of {nb}")
dur = (t[:, 2] - t[:, 5]) / 100
print(f"gate-open -> end per block: min {dur.min():.1f} median {np.median(dur):.1f} max {dur.max():.1f} us; last end {(t[:, 2].max() - t0) / 100:.1f} us")
rb = (C.c_longlong * (16 * nb))()
lib.kryst_debug_tw_rounds.argtypes = [C.POINTER(C.c_longlong), C.c_int32]
assert lib.kryst_debug_tw_rounds(rb, 16 * nb) == 0
rr = np.array(rb, dtype=np.int64).reshape(nb, 8, 2)
print("poller rounds (end time us : rows delivered / asked next):")
for b in list(range(min(nb, 4))) + ([nb - 1] if nb > 4 else []):
    print(f"  block {b:4d}: " + "  ".join(f"{(rr[b, i, 0] - t0) / 100:7.1f}:{rr[b, i, 1] // 100}/{rr[b, i, 1] % 100}" for i in range(8)))

sb = (C.c_longlong * (64 * nb))()
lib.kryst_debug_tw_steps.argtypes = [C.POINTER(C.c_longlong), C.c_int32]
assert lib.kryst_debug_tw_steps(sb, 64 * nb) == 0
ss = (np.array(sb, dtype=np.int64).reshape(nb, 64) - t0) / 100.0
print("block b (west neighbour b-1): step t published - producer finished step t+7 | consumer finished step t - published   (us)")
for b in range(1, min(nb, nbj, 5)):
    d1 = [ss[b, 32 + t] - ss[b - 1, t + 7] for t in range(0, 24)]
    d2 = [ss[b, t] - ss[b, 32 + t] for t in range(0, 24)]
    print(f"  block {b}: " + " ".join(f"{x:5.2f}" for x in d1))
    print(f"           " + " ".join(f"{x:5.2f}" for x in d2))
print("producer step times of block 0 (us): " + " ".join(f"{x:5.2f}" for x in ss[0, :32]))
