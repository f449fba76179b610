#!/usr/bin/env python3
"""Time kryst_csr_create (host CSR -> device, incl. validation and the three re-encodings).  usage: create_time.py [grid=128]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import kryst_amd as K
N = int(sys.argv[1]) if len(sys.argv) > 1 else 128
ctx = K.Context(0)
t0 = time.perf_counter(); rp, ci, va = K.host_stencil7(N, "convdiff"); t1 = time.perf_counter()
a = K.CsrMatrix.from_csr(N ** 3, N ** 3, rp, ci, va, ctx=ctx); t2 = time.perf_counter()
print(f"grid {N}: host generation {t1 - t0:.2f} s, kryst_csr_create {t2 - t1:.2f} s for {len(va) / 1e6:.1f} M nonzeros, encoding {a.encoding()}")
rng = np.random.default_rng(1)
va2 = rng.standard_normal(len(va))
t0 = time.perf_counter(); b = K.CsrMatrix.from_csr(N ** 3, N ** 3, rp, ci, va2, ctx=ctx); t1 = time.perf_counter()
print(f"grid {N}: random values: kryst_csr_create {t1 - t0:.2f} s, encoding {b.encoding()}")
