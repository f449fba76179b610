#!/usr/bin/env python3
"""Stress of the barrier-free run kernel (tri_run_free_kernel): true ILU(0) applies of random band matrices of several widths and row lengths, with
every wave count and loop form, NaNs in LDS first, against the oracle.   usage: free_stress.py [rows=120000] [seeds=3]"""
import itertools, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, scipy.sparse as sp
import kryst_amd as K
from oracle import oracle as O
NR = int(sys.argv[1]) if len(sys.argv) > 1 else 120000
SEEDS = int(sys.argv[2]) if len(sys.argv) > 2 else 3
ctx = K.Context(0)
bad = 0; runs = 0; t0 = time.time()
for seed, band, per in itertools.product(range(SEEDS), (12, 150, 1500), (4, 9, 17, 33)):
    rng = np.random.default_rng(1000 * seed + band + per)
    rows = np.repeat(np.arange(NR), per)
    cols = np.clip(rows + rng.integers(-band, band + 1, len(rows)), 0, NR - 1)
    far = rng.random(len(rows)) < 0.01                                  # one entry in a hundred anywhere: operands far outside the ring
    cols = np.where(far, rng.integers(0, NR, len(rows)), cols)
    m = sp.csr_matrix((rng.uniform(-1.0, 1.0, len(rows)), (rows, cols)), shape=(NR, NR)); m.sum_duplicates()
    m = (m - sp.diags(m.diagonal()) + sp.diags(np.asarray(abs(m).sum(axis=1)).ravel() + 1.0)).tocsr(); m.sort_indices()
    a_o = O.Csr(NR, NR, m.indptr, m.indices, m.data)
    ref = O.Pc.ilu0_true(a_o)
    a = K.CsrMatrix.from_csr(NR, NR, m.indptr, m.indices, m.data, ctx=ctx)
    r = rng.standard_normal(NR); zr = ref.apply(r)
    for waves, tune in itertools.product(("8", "4", "2"), ("16897", "513", "8705")):
        os.environ["KRYST_ILU_FREE_WAVES"] = waves; os.environ["KRYST_ILU_FREE_TUNE"] = tune; os.environ["KRYST_ILU_SYNCFREE"] = "0"
        pc = K.TrueIlu0().setup(a)
        ctx.poison_lds()
        for rep in range(2):
            z = pc.apply(r); runs += 1
            if not np.array_equal(z, zr):
                bad += 1
                d = np.flatnonzero(z != zr)
                print(f"MISMATCH seed={seed} band={band} per={per} waves={waves} tune={tune} rep={rep}: {len(d)} rows differ, first {d[:4]}, nan {int(np.isnan(z).sum())}", flush=True)
        del pc
    info = f"seed={seed} band={band} per={per}: ok so far ({runs} applies, {bad} bad, {time.time() - t0:.0f} s)"
    print(info, flush=True)
    del a
# 27-point operators on boxes through the level-ordered forms: every row a chain of two virtual rows, narrow dependency cones (waves can run ahead)
os.environ["KRYST_ILU_BOX"] = "0"
t = lambda n: sp.diags([np.ones(n - 1), np.ones(n), np.ones(n - 1)], [-1, 0, 1])
for seed, (Ni, Nj, Nk) in itertools.product(range(SEEDS), ((41, 30, 19), (64, 20, 15), (100, 12, 9), (30, 30, 30), (200, 9, 5))):
    rng = np.random.default_rng(77 + seed)
    m = (sp.identity(Ni * Nj * Nk) * 28.0 - sp.kron(t(Nk), sp.kron(t(Nj), t(Ni)))).tocsr(); m.sort_indices()
    m.data = m.data * rng.uniform(0.5, 1.5, len(m.data))
    n = m.shape[0]
    a_o = O.Csr(n, n, m.indptr, m.indices, m.data); ref = O.Pc.ilu0_true(a_o)
    a = K.CsrMatrix.from_csr(n, n, m.indptr, m.indices, m.data, ctx=ctx)
    r = rng.standard_normal(n); zr = ref.apply(r)
    for waves, tune in itertools.product(("8", "4", "2"), ("16897", "513", "8705")):
        os.environ["KRYST_ILU_FREE_WAVES"] = waves; os.environ["KRYST_ILU_FREE_TUNE"] = tune
        pc = K.TrueIlu0().setup(a)
        assert pc.ilu_info()["form"].startswith("level"), pc.ilu_info()
        ctx.poison_lds()
        for rep in range(3):
            z = pc.apply(r); runs += 1
            if not np.array_equal(z, zr):
                bad += 1
                print(f"MISMATCH box {Ni}x{Nj}x{Nk} seed={seed} waves={waves} tune={tune} rep={rep}: {int((z != zr).sum())} rows differ, nan {int(np.isnan(z).sum())}", flush=True)
        del pc
    print(f"box {Ni}x{Nj}x{Nk} seed={seed}: ok so far ({runs} applies, {bad} bad, {time.time() - t0:.0f} s)", flush=True)
    del a
print("STRESS_OK" if bad == 0 else f"STRESS_FAILED: {bad} of {runs}")
