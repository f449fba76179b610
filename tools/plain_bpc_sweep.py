"""Plain-CSR SpMV (spmv_wave_kernel) at grid^3 against the number of resident workgroups per CU (KRYST_SPMV_BLOCKS_PER_CU: a persistent grid of
that many workgroups per CU striding over the tiles; 0 = one tile per workgroup), window slots and tile order.  usage: plain_bpc_sweep.py [grid=512]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kryst_amd as K
grid = int(sys.argv[1]) if len(sys.argv) > 1 else 512
os.environ["KRYST_SPMV_COMPRESS"] = "0"
ctx = K.Context(0)
a = K.CsrMatrix.stencil7(grid, "poisson", ctx=ctx)
n = a.nrows(); x = ctx.vec(n).fill_splitmix(5); y = ctx.vec(n)
alg = 12 * a.nnz + 4 * (n + 1) + 16 * n
def t(env):
    for k, v in env.items(): os.environ[k] = v
    ms = sorted(a.bench_spmv(x, y, fused_dots=1, reps=10) for _ in range(3))[1]
    for k in env: os.environ.pop(k, None)
    return ms
base = t({})
print(f"{grid}^3 default: {base:.4f} ms ({alg / base / 8e9:.3f})", flush=True)
for bpc in (2, 3, 4, 5, 6, 8, 12, 16):
    for order in ("1", "0"):
        ms = t({"KRYST_SPMV_BLOCKS_PER_CU": str(bpc), "KRYST_SPMV_ORDER": order})
        print(f"  blocks/CU {bpc:2d} order {order}: {ms:.4f} ms ({alg / ms / 8e9:.3f})", flush=True)
for slots in ("2", "4", "7"):
    ms = t({"KRYST_SPMV_SLOTS": slots})
    print(f"  slots {slots}: {ms:.4f} ms ({alg / ms / 8e9:.3f})", flush=True)
print(f"skeleton {a.bench_csr_skeleton(x, y, reps=10):.4f} ms", a.placement_info())
