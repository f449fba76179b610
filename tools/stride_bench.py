#!/usr/bin/env python3
"""How does the byte distance between the vectors of a multi-stream BLAS-1 kernel affect its speed?
usage: stride_bench.py [grid=256]   (vectors of grid^3 doubles; prints GB/s per kind and stride offset)"""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kryst_amd as K
from kryst_amd._ffi import lib, check

N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
n = N ** 3
ctx = K.Context(0)
base = (n + 511) // 512 * 512 * 8 + 4096
words = {0: 4, 1: 9, 2: 6}
extras = [0, 256, 1024, 4096, 4096 + 256, 16384, 65536, 65536 + 4096, 1 << 20, (1 << 20) + 4096 + 256, (2 << 20) - (base % (2 << 20)),
          (2 << 20) - (base % (2 << 20)) + 4096 + 256, (2 << 20) - (base % (2 << 20)) + 65536 * 3 + 4096]
for kind in (0, 1, 2):
    for rnd in range(2):
        row = {}
        for ex in extras:
            ms = C.c_double(0)
            check(lib().kryst_bench_streams(ctx.h, n, base + ex, kind, 20, C.byref(ms)))
            row[ex] = round(words[kind] * n * 8 / ms.value / 1e6)
        print(json.dumps({"kind": kind, "round": rnd, "GBs_by_extra_stride": row}), flush=True)
