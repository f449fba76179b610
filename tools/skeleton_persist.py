"""Experiment: the plain-CSR traffic skeleton (one tile per workgroup) against the PERSISTENT software-pipelined one (one workgroup per CU striding over the
tiles, the next two tiles' matrix streams in flight by LDS-DMA; bench_streams.hip: csr_skeleton_persist_kernel).  usage: skeleton_persist.py [grid=512] [rounds=3] [grids=256,192,128]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kryst_amd as K
from kryst_amd._ffi import lib, check, Handle, c_dp
grid = int(sys.argv[1]) if len(sys.argv) > 1 else 512
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
wgs = [int(v) for v in (sys.argv[3] if len(sys.argv) > 3 else "256,192,128").split(",")]
L = lib()
L.kryst_debug_csr_skeleton_persist.restype = C.c_int32
L.kryst_debug_csr_skeleton_persist.argtypes = [Handle, Handle, Handle, C.c_int32, C.c_int32, C.c_int32, c_dp]
ctx = K.Context(0)
os.environ["KRYST_SPMV_COMPRESS"] = "0"
a = K.CsrMatrix.stencil7(grid, "poisson", ctx=ctx)
n = a.nrows(); x = ctx.vec(n).fill(1.0); y = ctx.vec(n)
alg = 12 * a.nnz + 4 * (n + 1) + 16 * n
for r in range(rounds):
    sk = a.bench_csr_skeleton(x, y, reps=10)
    out = [f"{grid}^3 round {r}: skeleton {sk:.4f} ms ({alg / sk / 8e9:.3f})"]
    for g in wgs:
        for aux in (0, 4, 8):
            ms = C.c_double(0); check(L.kryst_debug_csr_skeleton_persist(a.h, x.h, y.h, 10, g, aux, C.byref(ms)))
            out.append(f"persist grid {g} aux {aux}: {ms.value:.4f} ms ({alg / ms.value / 8e9:.3f})")
    kern = a.bench_spmv(x, y, fused_dots=1, reps=10)
    out.append(f"kernel {kern:.4f} ms ({alg / kern / 8e9:.3f})")
    print("   ".join(out), flush=True)
