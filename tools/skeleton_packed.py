"""Experiment: the plain-CSR traffic skeleton on the operator's three arrays against the same bytes as ONE chunked stream (256 columns + 256 values
per 3 KiB chunk).  usage: skeleton_packed.py [grid=512] [rounds=4]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kryst_amd as K
from kryst_amd._ffi import lib, check, Handle, c_dp
grid = int(sys.argv[1]) if len(sys.argv) > 1 else 512
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 4
L = lib()
L.kryst_debug_csr_skeleton_packed.restype = C.c_int32
L.kryst_debug_csr_skeleton_packed.argtypes = [Handle, Handle, Handle, C.c_int32, c_dp]
ctx = K.Context(0)
os.environ["KRYST_SPMV_COMPRESS"] = "0"
a = K.CsrMatrix.stencil7(grid, "poisson", ctx=ctx)
n = a.nrows(); x = ctx.vec(n).fill(1.0); y = ctx.vec(n)
alg = 12 * a.nnz + 4 * (n + 1) + 16 * n
for r in range(rounds):
    sk = a.bench_csr_skeleton(x, y, reps=10)
    ms = C.c_double(0); check(L.kryst_debug_csr_skeleton_packed(a.h, x.h, y.h, 10, C.byref(ms)))
    kern = a.bench_spmv(x, y, fused_dots=1, reps=10)
    print(f"{grid}^3 round {r}: skeleton {sk:.4f} ms ({alg / sk / 8e9:.3f})   packed skeleton {ms.value:.4f} ms ({alg / ms.value / 8e9:.3f})   kernel {kern:.4f} ms ({alg / kern / 8e9:.3f})", flush=True)
print(a.placement_info())
