#!/usr/bin/env python3
"""CG x/r update stream (4 reads, 2 writes) with plain / nontemporal loads and stores.  usage: nt_bench.py [grid=512]"""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kryst_amd as K
from kryst_amd._ffi import lib, check
N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
n = N ** 3
ctx = K.Context(0)
base = (n + 511) // 512 * 512 * 8 + 4096
for rnd in range(3):
    row = {}
    for kind, name in ((2, "plain"), (3, "nt_ld_st"), (4, "nt_st"), (5, "nt_ld")):
        for bpc in ("2", "4"):
            os.environ["KRYST_EW_BLOCKS_PER_CU"] = bpc
            ms = C.c_double(0)
            check(lib().kryst_bench_streams(ctx.h, n, base, kind, 20, C.byref(ms)))
            row[f"{name}/bpc{bpc}"] = round(6 * n * 8 / ms.value / 1e6)
    print(json.dumps(row), flush=True)
