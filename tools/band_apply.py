#!/usr/bin/env python3
"""Apply time of true ILU(0) on the random band matrix of tools/ilu_general.py (2 M rows, 9 per row, |i - j| <= 2000: 10 716 narrow dependency
levels per factor, the one-workgroup run kernel), for A/B runs of its settings.   usage: band_apply.py [rows=2000000] [BAND=2000] [PER=9] [KNOB=value[,KNOB=value] ...]
(BAND=w: |i - j| <= w instead of 2000 -- narrower bands give deeper, narrower level structures)"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import scipy.sparse as sp
import kryst_amd as K

NR = int(sys.argv[1]) if len(sys.argv) > 1 else 2000000
BAND = 2000
PER = 9
if len(sys.argv) > 2 and sys.argv[2].startswith("BAND="):
    BAND = int(sys.argv.pop(2)[5:])
if len(sys.argv) > 2 and sys.argv[2].startswith("PER="):
    PER = int(sys.argv.pop(2)[4:])
ctx = K.Context(0)
rng = np.random.default_rng(1)
rows = np.repeat(np.arange(NR), PER)
cols = np.clip(rows + rng.integers(-BAND, BAND + 1, len(rows)), 0, NR - 1)
m = sp.csr_matrix((rng.uniform(-1.0, 1.0, len(rows)), (rows, cols)), shape=(NR, NR))
m.sum_duplicates()
m = m - sp.diags(m.diagonal()) + sp.diags(np.asarray(abs(m).sum(axis=1)).ravel() + 1.0)
m = m.tocsr(); m.sort_indices()
a = K.CsrMatrix.from_csr(NR, NR, m.indptr, m.indices, m.data, ctx=ctx)
r = ctx.vec(NR).fill_splitmix(3); z = ctx.vec(NR)
for knobs in [{}] + [dict(kv.split("=") for kv in arg.split(",")) for arg in sys.argv[2:]]:
    for k, v in knobs.items():
        os.environ[k] = v
    pc = K.TrueIlu0().setup(a)                    # (a fresh preconditioner per setting: the apply's launch sequence is captured in a graph at its first use)
    info = pc.ilu_info()
    ms = min(pc.bench_apply(r, z, 5) for _ in range(2))
    print(json.dumps({"rows": NR, "band": BAND, "per_row": PER, "knobs": knobs, "apply_ms": ms, "levels": info["levels"], "us_per_level": ms * 1e3 / sum(info["levels"])}), flush=True)
    del pc
    for k in knobs:
        del os.environ[k]
