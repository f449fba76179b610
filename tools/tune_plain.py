#!/usr/bin/env python3
"""Plain-CSR SpMV (spmv_wave_kernel / spmv_rows_kernel without codes): tile->XCD map, nontemporal streams, window size.
One process, interleaved rounds.  usage: tune_plain.py [grid] [rounds]"""
import os, sys, statistics, itertools
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kryst_amd as K

grid = int(sys.argv[1]) if len(sys.argv) > 1 else 512
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
ctx = K.Context(0)
a = K.CsrMatrix.stencil7(grid, "poisson", ctx=ctx)
n = a.nrows()
x = ctx.vec(n).fill_splitmix(0xC0FFEE)
y = ctx.vec(n)
b = 12 * a.nnz + 4 * (n + 1) + 16 * n
os.environ["KRYST_SPMV_COMPRESS"] = "0"
keys = ("KRYST_SPMV_KERNEL", "KRYST_SPMV_NT", "KRYST_SPMV_SLOTS", "KRYST_SPMV_SWIZZLE", "KRYST_SPMV_GROUP", "KRYST_SPMV_BLOCKS_PER_CU", "KRYST_SPMV_PIPE_TPW", "KRYST_SPMV_MINW")
configs = []
mode = sys.argv[3] if len(sys.argv) > 3 else "pipe"
if mode == "tile":
    configs.append(("2", "0", "4", "0", "1", "0", "8", "1"))
    configs.append(("2", "0", "7", "0", "1", "0", "8", "1"))
    for nt, slots, minw, grp in itertools.product(("0", "1"), ("2", "4", "7"), ("1", "8"), ("1", "4")):
        if slots == "7" and minw == "8": continue
        configs.append(("5", nt, slots, "0", grp, "0", "8", minw))
elif mode == "wave":
    for nt, slots, grp in itertools.product(("0", "1"), ("2", "4", "7"), ("1", "8")):
        configs.append(("2", nt, slots, "0", grp, "0", "8", "1"))
else:
    configs.append(("2", "0", "4", "0", "1", "0", "8", "1"))       # the un-pipelined kernel, its two best settings
    configs.append(("2", "0", "7", "0", "1", "0", "8", "1"))
    for nt, slots, tpw in itertools.product(("0", "1"), ("2", "4", "7"), ("2", "4", "8")):
        configs.append(("4", nt, slots, "0", "1", "0", tpw, "1"))
res = {c: [] for c in configs}
for r in range(rounds):
    for c in configs:
        for k, v in zip(keys, c):
            os.environ[k] = v
        res[c].append(a.bench_spmv(x, y, fused_dots=1, reps=10))
print(f"grid {grid}: kern nt slots swz grp bpc tpw minw   median_ms  min_ms   GB/s(algorithmic)  frac_of_8TB/s")
for c in sorted(configs, key=lambda c: statistics.median(res[c])):
    med, mn = statistics.median(res[c]), min(res[c])
    print("   " + " ".join(f"{v:>4s}" for v in c) + f"   {med:8.4f} {mn:8.4f}   {b / med / 1e6:9.1f}   {b / med / 1e6 / 8000:.3f}", flush=True)
