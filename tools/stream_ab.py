#!/usr/bin/env python3
"""In-process A/B of the BLAS-1 stream shapes (kryst_bench_streams kinds) against KRYST_EW_BLOCKS_PER_CU (read per launch): the
settings are timed in turn, several rounds, median per setting.
usage: stream_ab.py [grid=512] [rounds=5] [kinds=2,6,7,8] [bpcs=0,2,3,4,6,8]"""
import ctypes as C, os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kryst_amd as K
from kryst_amd._ffi import lib, check

grid = int(sys.argv[1]) if len(sys.argv) > 1 else 512
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 5
kinds = [int(k) for k in (sys.argv[3] if len(sys.argv) > 3 else "2,6,7,8").split(",")]
bpcs = [int(k) for k in (sys.argv[4] if len(sys.argv) > 4 else "0,2,3,4,6,8").split(",")]
WORDS = {0: 4, 1: 9, 2: 6, 6: 3, 7: 3, 8: 5}
NAME = {0: "link 3R1W+dot", 1: "8 dots 9R", 2: "CgUpdate1 4R2W+dot", 6: "Aypx 2R1W", 7: "CgResidual 2R1W+dot", 8: "CgDirection 3R2W"}
ctx = K.Context(0)
n = grid ** 3
stride = ((n + 511) // 512 * 512 + 512) * 8
t = {(k, b): [] for k in kinds for b in bpcs}
for r in range(rounds):
    for k in kinds:
        for b in bpcs:
            if b: os.environ["KRYST_EW_BLOCKS_PER_CU"] = str(b)
            else: os.environ.pop("KRYST_EW_BLOCKS_PER_CU", None)
            ms = C.c_double(0)
            check(lib().kryst_bench_streams(ctx.h, n, stride, k, 20, C.byref(ms)))
            t[(k, b)].append(ms.value)
for k in kinds:
    for b in bpcs:
        med = statistics.median(t[(k, b)])
        print(f"grid {grid} {NAME[k]:22s} bpc {b or 'default'}: {med:.4f} ms  {WORDS[k] * 8 * n / med / 1e6 / 8000:.3f} of 8 TB/s", flush=True)
