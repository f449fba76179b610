#!/usr/bin/env python3
"""bench.py -- CG iterations/sec + SpMV GB/s against the HBM roofline on synthetic 3-D 7-point Poisson CSR.

    python bench.py --gpus N --steps K --warmup W            (N > 1: launched by torch.distributed.run)

A "step" is ONE iteration of the reference's unpreconditioned CG (src/solver/cg.rs:141-285: SpMV + 2 inner
products + 3 vector updates) on a matrix, right-hand side and iterate that are already resident in HBM.
W warm-up iterations and exactly K timed iterations run inside one stepping session (tol = 0, so the device
never stops early); the timed region is bracketed by barrier + device synchronize on both sides and the maximum
over ranks is reported.  One JSON line on rank 0.

Workload: ONE fixed 512^3 system for every N (BASELINE.json's metric is quoted on 256^3/512^3; north_star asks for
strong scaling of CG iterations/sec on 512^3, which fits a single 288 GB GPU).  N > 1: row-partitioned in k-slabs over
the ranks, halo planes over xGMI, inner products by RCCL all-gather.  `--grid 256` measures configs[1]'s grid instead
(profiles/r01/bench_256.json); `--with-256` adds it to the same line as "config1_256".
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0           # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (6.29 measured copy)


def spmv_bytes(n, nnz):
    """Algorithmic bytes of one CSR SpMV: 12 B per nnz (f64 value + i32 column), 4 B per row pointer,
    x read once and y written once (16 B per row)."""
    return 12 * nnz + 4 * (n + 1) + 16 * n


def cpu_baseline(grid, seconds=12.0):
    """The oracle's CG (the CPU restatement of the reference path, OpenMP over rows / tiles like the reference's
    Rayon loops) timed on the host cores on a BOUNDED sample of the same workload: CG on the 256^3 Poisson system
    (1/8 of the 512^3 rows when grid = 512; every pass is a bandwidth-bound stream, so a 512^3 iteration costs 8x),
    as many iterations as fit in ~`seconds`."""
    import numpy as np
    import kryst_amd as K
    from oracle import oracle as O
    cores = min(len(os.sched_getaffinity(0)), 16)      # the GPU box's CPU share for one GPU
    O.set_threads(cores)
    T, V, F = K.reduce_spec()
    sgrid = min(grid, 256)
    rp, ci, va = K.host_stencil7(sgrid, "poisson")
    a = O.Csr(sgrid ** 3, sgrid ** 3, rp, ci, va, check=False)
    b = a.spmv(np.ones(a.nrows))
    rs = O.Reduce.tiled(T, V, F)
    t0 = time.perf_counter(); O.solve("cg", a, b, tol=0.0, max_iters=3, rs=rs); t3 = time.perf_counter() - t0
    iters = int(max(5, min(400, seconds / max(t3 / 4.0, 1e-4))))     # 3 iterations + the initial residual pass
    t0 = time.perf_counter()
    res = O.solve("cg", a, b, tol=0.0, max_iters=iters, rs=rs)
    dt = time.perf_counter() - t0
    scale = (sgrid / grid) ** 3
    # the bit-canonical single-thread figure beside it, a few iterations only
    O.set_threads(1)
    t0 = time.perf_counter(); r1 = O.solve("cg", a, b, tol=0.0, max_iters=4, rs=rs); dt1 = time.perf_counter() - t0
    O.set_threads(cores)
    return {"value": res.iterations / dt * scale, "unit": "cg_iterations/s", "cores": cores, "kind": "port",
            "single_thread_value": r1.iterations / dt1 * scale,
            "sample": f"{res.iterations} oracle CG iterations on a {sgrid}^3 Poisson system in {dt:.1f} s "
                      f"({res.iterations / dt:.1f} it/s, OpenMP rows/tiles over {cores} threads, device-order dot)"
                      + (f"; scaled by {scale:.4f} = ({sgrid}/{grid})^3 rows to the {grid}^3 workload" if scale != 1 else "")}


def traffic_of(grid):
    """HBM bytes per SpMV launch from the committed rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (tools/pmc_traffic.py,
    read side calibrated on a kernel of known byte count as MI355X_MICROARCH.md prescribes), or None."""
    tf = os.path.join(ROOT, "profiles", "spmv_traffic.json")
    try:
        return json.load(open(tf))[str(grid)]["hbm_bytes_per_launch"]
    except Exception:
        return None


def roofline(enc, nloc, nnz_loc, spmv_ms, plain_ms, copy_gbs):
    """`achieved` prices the launch at the ALGORITHMIC CSR bytes of SURVEY 8(d) (12 B/nnz + row pointers + x + y) whatever
    the operator's storage form; `bytes_moved_model` is what the kernel that ran actually streams (a lossless re-encoding
    moves fewer bytes than the CSR arrays, so frac can exceed 1); `plain_csr` is the same launch on the CSR arrays."""
    name, npat, ntab = enc
    alg = spmv_bytes(nloc, nnz_loc)
    moved = {"csr": alg, "csr-d8": alg - 3 * nnz_loc, "csr-d16": alg - 10 * nnz_loc,
             "csr-p16": 2 * nloc + 16 * nloc}[name]
    kern = {"csr": "spmv_wave_kernel<1> (plain CSR: 8 B value + 4 B column per entry)",
            "csr-d8": "spmv_rows_kernel<1> (CSR-D8: 8 B value + 1-byte column-offset code per entry)",
            "csr-d16": "spmv_dict_kernel<1> (CSR-D16: one 16-bit word per entry = offset code + value code)",
            "csr-p16": f"spmv_pattern_kernel<1> (CSR-P16: one 16-bit row-pattern id per row; {npat} ids, {ntab} table entries in LDS)"}[name]
    ach = alg / (spmv_ms * 1e-3) / 1e9
    return {"bound": "hbm", "kernel": kern + ", fused (p,Ap) partials; achieved = ALGORITHMIC CSR bytes (12 B/nnz) / time",
            "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
            "bytes_per_launch": alg, "ms_per_launch": spmv_ms, "traffic": None,
            "encoding": name, "bytes_moved_model": moved, "moved_GBs": moved / (spmv_ms * 1e-3) / 1e9,
            "plain_csr": {"ms_per_launch": plain_ms, "achieved": alg / (plain_ms * 1e-3) / 1e9,
                          "frac": alg / (plain_ms * 1e-3) / 1e9 / HBM_PEAK_GBS},
            "measured_copy_GBs": copy_gbs}


def blas1_streams(K, ctx, n):
    """The two BLAS-1 kernel shapes of a CG iteration, timed live with HIP events on fresh vectors of this size
    (kryst_bench_streams): algorithmic bytes = 48 n (x += a p, r -= a Ap, (r,r): 4 reads + 2 writes) and 24 n (p = r + b p)."""
    import ctypes as C
    from kryst_amd._ffi import lib, check
    stride = ((n + 511) // 512 * 512 + 512) * 8
    out = []
    for kind, name, words in ((2, "ew_kernel<CgUpdate1> (x += alpha p, r -= alpha Ap, fused (r,r))", 6), (6, "ew_kernel<AypxDevOp> (p = r + beta p)", 3)):
        ms = C.c_double(0)
        check(lib().kryst_bench_streams(ctx.h, n, stride, kind, 20, C.byref(ms)))
        ach = words * 8 * n / (ms.value * 1e-3) / 1e9
        out.append({"kernel": name, "bound": "hbm", "bytes_per_launch": words * 8 * n, "ms_per_launch": ms.value,
                    "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS})
    return out


def run_cg(K, ctx, dist, grid, solver, warmup, steps):
    """W warm-up + exactly K timed iterations of one stepping session; returns (seconds, stats, spmv_ms, a)."""
    def barrier():
        ctx.synchronize()
        if dist is not None:
            dist.barrier()

    a = K.CsrMatrix.stencil7(grid, "poisson", ctx=ctx)
    nloc = a.nrows()
    ones = ctx.vec(nloc).fill(1.0)
    b = a.spmv(ones)                      # b = A*1 (tests/preconditioner_integration.rs:25-31 convention)
    x = ctx.vec(nloc)
    pc = K.Jacobi().setup(a) if solver == "pcg" else None
    sess = K.Session(solver, a, pc, b, x, tol=0.0, max_iters=warmup + steps)
    sess.step(warmup)
    barrier()
    t0 = time.perf_counter()
    sess.step(steps)
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        import torch
        t = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t[0])
    stats = sess.end()
    assert stats.iterations == warmup + steps, stats
    # dominant kernel: the SpMV with the fused (p,Ap) partials, timed live with HIP events on its own stream
    y = ctx.vec(nloc)
    spmv_ms = a.bench_spmv(b, y, fused_dots=1, reps=50)
    enc = a.encoding()
    # the same launch on the plain CSR arrays (12 B/nnz streamed), for reference
    prev = os.environ.get("KRYST_SPMV_COMPRESS")
    os.environ["KRYST_SPMV_COMPRESS"] = "0"
    plain_ms = a.bench_spmv(b, y, fused_dots=1, reps=20)
    if prev is None:
        del os.environ["KRYST_SPMV_COMPRESS"]
    else:
        os.environ["KRYST_SPMV_COMPRESS"] = prev
    # context for the roofline: the device-copy rate at this footprint (hipMemcpy D2D of one vector, read + write)
    y.copy_from(b); ctx.synchronize()
    ctx.timer_start()
    for _ in range(10):
        y.copy_from(b)
    copy_gbs = 10 * 16.0 * nloc / (ctx.timer_stop() * 1e-3) / 1e9
    return dt, stats, spmv_ms, nloc, a.nnz, copy_gbs, enc, plain_ms, blas1_streams(K, ctx, nloc)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--grid", type=int, default=0, help="grid edge (default 512 for every N)")
    ap.add_argument("--solver", default="cg", choices=["cg", "pcg"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--with-256", action="store_true",
                    help="also measure BASELINE configs[1]'s 256^3 grid in the same process (block `config1_256`); off by "
                         "default so that the rocprofv3 averages of the default command refer to ONE problem size")
    args = ap.parse_args()

    # a hung collective must not hang the node: give up loudly after 20 minutes
    import threading
    def _watchdog():
        sys.stderr.write("bench.py: watchdog timeout (1200 s), aborting\n"); sys.stderr.flush(); os._exit(124)
    wd = threading.Timer(1200.0, _watchdog); wd.daemon = True; wd.start()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N with N > 1 must be launched by torch.distributed.run (one rank per GPU)")
        args.gpus = world
    grid = args.grid or 512          # ONE fixed problem for every N (strong scaling, north_star): 512^3

    dist = None
    if world > 1:
        # torch.distributed (gloo) is plumbing only: ship the RCCL unique id and provide the host barrier / max
        import torch.distributed as dist
        dist.init_process_group("gloo", rank=rank, world_size=world)
    import kryst_amd as K

    if world > 1:
        box = [K.Context.unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        # KRYST_BENCH_DEVICE: rehearsal of the N > 1 flow on a one-GPU box (all ranks on one device, RCCL stand-in)
        dev = int(os.environ.get("KRYST_BENCH_DEVICE", local_rank))
        ctx = K.Context(dev, rank, world, box[0])
    else:
        ctx = K.Context(0)

    n = grid ** 3
    nnz = 7 * n - 6 * grid * grid
    dt, stats, spmv_ms, nloc, nnz_loc, copy_gbs, enc, plain_ms, streams = run_cg(K, ctx, dist, grid, args.solver, args.warmup, args.steps)
    bytes_local = spmv_bytes(nloc, nnz_loc)
    achieved = bytes_local / (spmv_ms * 1e-3) / 1e9

    out = {
        "metric": "cg_iterations_per_sec", "value": args.steps / dt, "unit": "iterations/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"{'jacobi_pcg' if args.solver == 'pcg' else 'cg'}_poisson7_{grid}^3", "grid": grid,
                   "rows": n, "nnz": nnz, "partition": f"{world} k-slab(s)", "rhs": "A*1", "x0": "0",
                   "final_residual": stats.final_residual},
        "roofline": roofline(enc, nloc, nnz_loc, spmv_ms, plain_ms, copy_gbs),
        "roofline_blas1": streams,      # the other two kernels of the iteration (they take 60 % of it once the SpMV streams 18 B/row)
    }
    if world == 1:
        out["roofline"]["traffic"] = traffic_of(grid)
    if world == 1 and grid != 256 and args.with_256:
        dt2, st2, ms2, nl2, nz2, cp2, enc2, pm2, _s2 = run_cg(K, ctx, None, 256, args.solver, args.warmup, args.steps)
        out["config1_256"] = {"workload": f"{'jacobi_pcg' if args.solver == 'pcg' else 'cg'}_poisson7_256^3",
                              "value": args.steps / dt2, "unit": "iterations/s", "ms_per_step": dt2 / args.steps * 1e3,
                              "roofline": roofline(enc2, nl2, nz2, ms2, pm2, cp2)}
        tr2 = traffic_of(256)
        if tr2:
            out["config1_256"]["roofline"]["traffic"] = tr2
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            out["cpu_baseline"] = cpu_baseline(grid)
        except Exception as e:                      # the oracle is only the reported baseline, never the product
            out["cpu_baseline"] = {"value": None, "unit": "cg_iterations/s", "cores": 0, "kind": "port", "sample": f"failed: {e}"}
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
