#!/usr/bin/env python3
"""bench.py -- CG iterations/sec + SpMV GB/s against the HBM roofline on synthetic 3-D 7-point Poisson CSR.

    python bench.py --gpus N --steps K --warmup W            (N > 1: launched by torch.distributed.run, one rank per GPU)

A "step" is ONE iteration of the reference's unpreconditioned CG (src/solver/cg.rs:141-285: SpMV + 2 inner products + 3
vector updates) on a matrix, right-hand side and iterate that are already resident in HBM.  W warm-up iterations and exactly
K timed iterations run inside one stepping session (tol = 0, so the device never stops early); the timed region is bracketed
by barrier + device synchronize on both sides and the maximum over ranks is reported.  One JSON line on rank 0.

Workload: ONE fixed 512^3 system for every N (BASELINE.json's metric is quoted on 256^3 / 512^3; north_star asks for strong
scaling of CG iterations/sec on 512^3, which fits a single 288 GB GPU).  N > 1: row-partitioned in k-slabs over the ranks, halo
planes over xGMI, inner products by RCCL all-gather.

What the line holds (every `frac` is bytes the named kernel really moves / its HIP-event time / 8 TB/s, so it is <= 1):
  value             CG iterations/s with the operator in its default (most compact lossless) storage form
  value_plain_csr   the same K iterations with KRYST_SPMV_COMPRESS=0: the 12-bytes-per-entry CSR arrays are streamed, which is
                    what every matrix that is not a constant-coefficient stencil gets
  roofline          the SpMV kernel of the timed loop (fused (p,Ap) partials): bytes it moves (model; PMC `traffic` beside it --
                    at N = 1 measured by this very command for the 512^3 forms, two rocprofv3 --pmc passes per form in child
                    processes; otherwise from profiles/spmv_traffic.json when that was measured on this very source tree;
                    `traffic_source` says which), `algorithmic_*` = SURVEY 8(d)'s CSR bytes for comparison (a re-encoded
                    operator moves fewer)
  roofline_csr      the plain-CSR kernel on SURVEY 8(d)'s bytes: north_star's "% of HBM roofline on CSR SpMV"
  roofline_blas1    the two vector kernels of a CG iteration
  phase_ms          device time per iteration by phase (hipEvents between the phases, a separate short run), per rank
  config1_256       (N = 1) the same measurements on BASELINE configs[1]'s 256^3 grid, with the CPU port timed on that grid
  config3_gmres30_jacobi_256, config5_bicgstab_ilu0_256, config4_jacobi_pcg_512
                    (N = 1) the other BASELINE configs on one GPU: iterations/s of the solve to the config's tolerance (configs 3, 5)
                    or of a fixed number of stepped iterations (config 4), each with the roofline block of ITS dominant kernel
                    (Gram-Schmidt link; triangular solve -- priced at the bytes it moves, SURVEY 8(d)'s bytes labelled beside it)
  variable_coefficient_256 / _512
                    (N = 1) a 7-point operator with per-edge random coefficients (kind "varcoef"): neither CSR-P16 / D16 nor the
                    triangular solve's chunk dedup apply -- SpMV form / ms / fraction, CG iterations/s, ILU(0) apply ms
  cpu_baseline      the oracle's CG timed on the host cores AT the workload's size (512^3 when host memory allows, else the
                    256^3 sample scaled and marked "extrapolated": true)
"""
import argparse
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0           # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (6.29 measured copy)


def spmv_bytes(n, nnz):
    """Algorithmic bytes of one CSR SpMV: 12 B per nnz (f64 value + i32 column), 4 B per row pointer,
    x read once and y written once (16 B per row)."""
    return 12 * nnz + 4 * (n + 1) + 16 * n


# ---------------------------------------------------------------------------------------------------------------- launcher
class TorchGroup:
    """torch.distributed (gloo) as plumbing: ships the RCCL unique id, host barrier, max, gather."""

    def __init__(self, rank, world):
        import torch.distributed as dist
        self.dist, self.rank, self.world = dist, rank, world
        dist.init_process_group("gloo", rank=rank, world_size=world)

    def broadcast_bytes(self, payload):
        box = [payload if self.rank == 0 else None]
        self.dist.broadcast_object_list(box, src=0)
        return box[0]

    def barrier(self):
        self.dist.barrier()

    def allreduce_max(self, x):
        import torch
        t = torch.tensor([x], dtype=torch.float64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t[0])

    def gather(self, obj):
        out = [None] * self.world
        self.dist.all_gather_object(out, obj)
        return out

    def close(self):
        self.dist.barrier()
        self.dist.destroy_process_group()


class SoloGroup:
    rank, world = 0, 1
    def broadcast_bytes(self, payload): return payload
    def barrier(self): pass
    def allreduce_max(self, x): return x
    def gather(self, obj): return [obj]
    def close(self): pass


def make_group(kind, rank, world):
    if world == 1:
        return SoloGroup()
    if kind == "auto":
        kind = os.environ.get("KRYST_LAUNCHER", "torch")
    if kind != "socket":
        try:
            import torch.distributed                        # noqa: F401  (the same answer on every rank of one image)
        except ImportError:
            kind = "socket"
    if kind == "socket":                                    # no torch anywhere: kryst_amd/launch.py (TCP rendezvous)
        from kryst_amd.launch import Rendezvous
        return Rendezvous.from_env()
    return TorchGroup(rank, world)


# ---------------------------------------------------------------------------------------------------------------- CPU port
def host_mem_available_gb():
    try:
        avail = None
        for line in open("/proc/meminfo"):
            if line.startswith("MemAvailable:"):
                avail = int(line.split()[1]) / 1e6
        for p in ("/sys/fs/cgroup/memory.max", "/sys/fs/cgroup/memory/memory.limit_in_bytes"):
            if os.path.exists(p):
                v = open(p).read().strip()
                if v.isdigit():
                    used = 0
                    for q in ("/sys/fs/cgroup/memory.current", "/sys/fs/cgroup/memory/memory.usage_in_bytes"):
                        if os.path.exists(q):
                            used = int(open(q).read().strip())
                    avail = min(avail, (int(v) - used) / 1e9) if avail is not None else (int(v) - used) / 1e9
        return avail or 0.0
    except Exception:
        return 0.0


def cpu_cg(grid, seconds):
    """The oracle's CG (the CPU restatement of the reference path, OpenMP over rows / tiles like the reference's Rayon loops,
    device-order dot) on the grid^3 Poisson system: as many iterations as fit in ~`seconds`.  -> dict."""
    import numpy as np
    import kryst_amd as K
    from oracle import oracle as O
    cores = min(len(os.sched_getaffinity(0)), 16)      # the GPU box's CPU share for one GPU
    O.set_threads(cores)
    T, V, F = K.reduce_spec()
    rp, ci, va = K.host_stencil7(grid, "poisson")
    a = O.Csr(grid ** 3, grid ** 3, rp, ci, va, check=False)
    b = a.spmv(np.ones(a.nrows))
    rs = O.Reduce.tiled(T, V, F)
    t0 = time.perf_counter(); O.solve("cg", a, b, tol=0.0, max_iters=2, rs=rs); t2 = time.perf_counter() - t0
    iters = int(max(3, min(400, seconds / max(t2 / 3.0, 1e-4))))     # 2 iterations + the initial residual pass
    t0 = time.perf_counter()
    res = O.solve("cg", a, b, tol=0.0, max_iters=iters, rs=rs)
    dt = time.perf_counter() - t0
    out = {"value": res.iterations / dt, "unit": "cg_iterations/s", "cores": cores, "kind": "port", "grid": grid, "extrapolated": False,
           "sample": f"{res.iterations} oracle CG iterations on the {grid}^3 Poisson system in {dt:.1f} s "
                     f"(OpenMP rows/tiles over {cores} threads, device-order dot; usize = int64 indices like the reference)"}
    if grid <= 256:                                     # the bit-canonical single-thread figure beside it, a few iterations
        O.set_threads(1)
        t0 = time.perf_counter(); r1 = O.solve("cg", a, b, tol=0.0, max_iters=4, rs=rs); dt1 = time.perf_counter() - t0
        O.set_threads(cores)
        out["single_thread_value"] = r1.iterations / dt1
    return out


def cpu_baseline(grid, base256):
    """Measured AT the workload's size when the host has the memory for it (512^3: 24 GB of int64-indexed CSR + vectors),
    otherwise the 256^3 measurement scaled by the row count and marked as extrapolated."""
    try:
        if grid <= 256:
            return base256 or cpu_cg(grid, 8.0)
        need = 16.0 * 7 * grid ** 3 / 1e9 + 8 * 8.0 * grid ** 3 / 1e9 + 6.0
        if os.environ.get("KRYST_BENCH_CPU_FULL", "1") != "0" and host_mem_available_gb() > need:
            return cpu_cg(grid, 10.0)
        b = base256 or cpu_cg(256, 8.0)
        scale = (256 / grid) ** 3
        out = dict(b, value=b["value"] * scale, grid=grid, extrapolated=True)
        out["sample"] = b["sample"] + f"; scaled by {scale:.4f} = (256/{grid})^3 rows (host memory below {need:.0f} GB)"
        if "single_thread_value" in out:
            out["single_thread_value"] = b["single_thread_value"] * scale
        return out
    except Exception as e:                              # the oracle is only the reported baseline, never the product
        return {"value": None, "unit": "cg_iterations/s", "cores": 0, "kind": "port", "sample": f"failed: {e}"}


# ---------------------------------------------------------------------------------------------------------------- GPU side
_LIVE_TRAFFIC = {}
LIVE_FORMS = {(512, "default"), (512, "plain")}          # measured in THIS run (about 7 s per pass); the other sizes / forms come from profiles/


def live_traffic(grid, form):
    """HBM bytes per SpMV launch measured in THIS run: two rocprofv3 --pmc passes (FETCH_SIZE, then WRITE_SIZE -- separate runs, no
    trace domains, as MI355X_MICROARCH.md prescribes) over tools/spmv_only.py in child processes; the read side is calibrated on the
    ew_kernel<DotOp> launches of the same process, whose byte count is known (the counter under-reports wide coalesced reads on
    gfx950).  None when rocprofv3 is missing, a pass fails or KRYST_BENCH_LIVE_TRAFFIC=0."""
    key = (grid, form)
    if key in _LIVE_TRAFFIC:
        return _LIVE_TRAFFIC[key]
    res = None
    if os.environ.get("KRYST_BENCH_LIVE_TRAFFIC", "1") != "0" and key in LIVE_FORMS:
        import collections, csv, glob, shutil, subprocess, tempfile
        exe = shutil.which("rocprofv3")
        tmp = tempfile.mkdtemp(prefix="kryst_pmc_", dir="/tmp") if exe else None
        try:
            env = dict(os.environ, TMPDIR="/tmp")
            env.pop("KRYST_SPMV_COMPRESS", None)
            if form == "plain":
                env["KRYST_SPMV_COMPRESS"] = "0"
            means = {}
            for counter in ("FETCH_SIZE", "WRITE_SIZE"):
                d = os.path.join(tmp, counter)
                r = subprocess.run([exe, "--pmc", counter, "--output-format", "csv", "-d", d, "-o", "p", "--", sys.executable,
                                    os.path.join(ROOT, "tools", "spmv_only.py"), str(grid), "5", "1", "varcoef" if form == "varcoef" else "poisson"],
                                   cwd="/tmp", env=env, capture_output=True, timeout=90)
                if r.returncode != 0:
                    raise RuntimeError(f"rocprofv3 --pmc {counter} failed")
                acc = collections.defaultdict(list)
                for row in csv.DictReader(open(glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True)[0])):
                    if row["Counter_Name"] == counter:
                        acc[row["Kernel_Name"]].append(float(row["Counter_Value"]))
                means[counter] = {k: sum(v) / len(v) for k, v in acc.items()}
            spmv = [k for k in means["FETCH_SIZE"] if "spmv" in k and "<1," in k][0]
            dotk = [k for k in means["FETCH_SIZE"] if "DotOp" in k][0]
            cal = (2 * grid ** 3 * 8) / (means["FETCH_SIZE"][dotk] * 1024.0)
            res = means["FETCH_SIZE"][spmv] * 1024.0 * cal + means["WRITE_SIZE"][spmv] * 1024.0
        except Exception as e:
            sys.stderr.write(f"bench.py: live traffic measurement ({grid}, {form}) skipped: {e}\n")
            res = None
            LIVE_FORMS.clear()                                  # one failed pass: no further attempts in this run (the committed profile remains)
        finally:
            if tmp:
                shutil.rmtree(tmp, ignore_errors=True)
    _LIVE_TRAFFIC[key] = res
    return res


def traffic_of(grid, form):
    """HBM bytes per SpMV launch from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes: measured in this run for the headline forms
    (live_traffic), else read from profiles/spmv_traffic.json (tools/profile_round.sh, the same passes) -- but only when that
    measurement was made on THIS source tree (same sha of the files the SpMV kernels are built from), else None."""
    from kryst_amd._ffi import spmv_source_sha16
    live = live_traffic(grid, form)
    if live is not None:
        return live
    try:
        e = json.load(open(os.path.join(ROOT, "profiles", "spmv_traffic.json")))[str(grid)][form]
        if e.get("source_sha16") != spmv_source_sha16():
            return None
        return e["hbm_bytes_per_launch"]
    except Exception:
        return None


def kernel_name(enc, staged=False):
    name, npat, ntab = enc
    if name == "csr-p16" and staged:
        return (f"spmv_pattern_stage_kernel<1> (CSR-P16: one 16-bit row-pattern id per row; {npat} ids, {ntab} table entries; the near operands of "
                "a run of 2 or 4 tiles out of an LDS window filled by LDS-DMA loads)")
    return {"csr": "spmv_wave_kernel<1> (plain CSR: 8 B value + 4 B column per entry)",
            "csr-d8": "spmv_rows_kernel<1> (CSR-D8: 8 B value + 1-byte column-offset code per entry)",
            "csr-d16": "spmv_dict_kernel<1> (CSR-D16: one 16-bit word per entry = offset code + value code)",
            "csr-p16": f"spmv_pattern_kernel<1> (CSR-P16: one 16-bit row-pattern id per row; {npat} ids, {ntab} table entries in LDS)",
            "csr-dia": f"spmv_dia_kernel<1> (CSR-DIA: one 8-byte value stream per diagonal, {npat} diagonals; no row pointers, no column codes)"}[name]


def roofline_of(enc, grid, nloc, nnz_loc, ms, world, traffic_form=None, staged=False):
    """The kernel that ran, priced at the bytes IT moves."""
    alg = spmv_bytes(nloc, nnz_loc)
    moved = {"csr": alg, "csr-d8": alg - 3 * nnz_loc, "csr-d16": alg - 10 * nnz_loc, "csr-p16": 2 * nloc + 16 * nloc,
             "csr-dia": 8 * enc[1] * nloc + 16 * nloc}[enc[0]]
    ach = moved / (ms * 1e-3) / 1e9
    out = {"bound": "hbm", "kernel": kernel_name(enc, staged) + ", fused (p,Ap) tile partials",
           "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
           "bytes_per_launch": moved, "bytes_model": "bytes this storage form streams: matrix description + x once + y once",
           "ms_per_launch": ms, "traffic": None, "encoding": enc[0],
           "algorithmic_bytes": alg, "algorithmic_GBs": alg / (ms * 1e-3) / 1e9, "algorithmic_speedup": alg / moved}
    if world == 1:
        tform = traffic_form or ("plain" if enc[0] == "csr" else "default")
        tr = traffic_of(grid, tform)
        if tr:
            out["traffic"] = tr
            out["traffic_source"] = ("rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes run by this command" if _LIVE_TRAFFIC.get((grid, tform))
                                     else "profiles/spmv_traffic.json (same passes, same source sha)")
            out["frac_traffic"] = tr / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS
    return out


def roofline_csr_of(grid, nloc, nnz_loc, ms, world):
    alg = spmv_bytes(nloc, nnz_loc)
    ach = alg / (ms * 1e-3) / 1e9
    out = {"bound": "hbm", "kernel": kernel_name(("csr", 0, 0)) + ", fused (p,Ap) tile partials",
           "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
           "bytes_per_launch": alg, "bytes_model": "SURVEY 8(d): 12 nnz + 4 (n + 1) + 16 n", "ms_per_launch": ms, "traffic": None}
    if world == 1:
        tr = traffic_of(grid, "plain")
        if tr:
            out["traffic"] = tr
            out["traffic_source"] = ("rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes run by this command" if _LIVE_TRAFFIC.get((grid, "plain"))
                                     else "profiles/spmv_traffic.json (same passes, same source sha)")
    return out


def blas1_streams(K, ctx, n):
    """The two BLAS-1 kernel shapes of a CG iteration, timed live with HIP events on fresh vectors of this size
    (kryst_bench_streams): the residual pass r -= a Ap with the fused (r,r) (2 reads + 1 write: 24 n bytes) and the direction pass
    x += a p, p = r + b p (3 reads + 2 writes: 40 n) -- the reference's x += alpha p (cg.rs:207-209) rides on the pass that reads p
    anyway, so p is read once per iteration (64 n bytes per iteration instead of 72 n; KRYST_CG_DEFER_X=0: 48 n + 24 n)."""
    import ctypes as C
    from kryst_amd._ffi import lib, check
    stride = ((n + 511) // 512 * 512 + 512) * 8
    out = []
    if os.environ.get("KRYST_CG_DEFER_X", "1") == "0":
        shapes = ((2, "ew_kernel<CgUpdate1> (x += alpha p, r -= alpha Ap, fused (r,r))", 6), (6, "ew_kernel<AypxDevOp> (p = r + beta p)", 3))
    else:
        shapes = ((7, "ew_kernel<CgResidualOp> (r -= alpha Ap, fused (r,r))", 3), (8, "ew_kernel<CgDirectionOp> (x += alpha p, p = r + beta p)", 5))
    for kind, name, words in shapes:
        ms = C.c_double(0)
        check(lib().kryst_bench_streams(ctx.h, n, stride, kind, 20, C.byref(ms)))
        ach = words * 8 * n / (ms.value * 1e-3) / 1e9
        out.append({"kernel": name, "bound": "hbm", "bytes_per_launch": words * 8 * n, "ms_per_launch": ms.value,
                    "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS})
    return out


class env_override:
    def __init__(self, **kv): self.kv, self.old = kv, {}
    def __enter__(self):
        for k, v in self.kv.items():
            self.old[k] = os.environ.get(k)
            os.environ[k] = v
    def __exit__(self, *a):
        for k, v in self.old.items():
            if v is None: os.environ.pop(k, None)
            else: os.environ[k] = v


def measure(K, ctx, group, grid, solver, warmup, steps, phase_iters):
    """Everything measured on one grid size -> dict (the operator is built once; the plain-CSR figures re-run the same
    iterations with KRYST_SPMV_COMPRESS=0, which the launcher reads per launch)."""
    def barrier():
        ctx.synchronize()
        group.barrier()

    world = group.world
    a = K.CsrMatrix.stencil7(grid, "poisson", ctx=ctx)
    nloc = a.nrows()
    b = a.spmv(ctx.vec(nloc).fill(1.0))                      # b = A*1 (tests/preconditioner_integration.rs:25-31 convention)
    pc = K.Jacobi().setup(a) if solver == "pcg" else None

    def timed_iterations():
        x = ctx.vec(nloc)
        sess = K.Session(solver, a, pc, b, x, tol=0.0, max_iters=warmup + steps)
        sess.step(warmup)
        barrier()
        t0 = time.perf_counter()
        sess.step(steps)
        barrier()
        dt = group.allreduce_max(time.perf_counter() - t0)
        stats = sess.end()
        assert stats.iterations == warmup + steps, stats
        return dt, stats

    dt, stats = timed_iterations()
    # N > 1: the same K iterations with the inner products crossing the ranks through the hipIpc mailboxes instead of two RCCL
    # all-gathers per iteration (kryst_ctx_scalar_reduce).  The faster path that reproduces the other's residual BIT FOR BIT is the
    # headline; both figures are reported.  (Every rank takes the same decision: times are max-reduced, scalars are identical.)
    reduce_info = None
    if world > 1:
        reduce_info = {"path": "rccl", "value_rccl": steps / dt, "value_ipc": None, "ipc": "unavailable on this node (hipIpc export / mapping failed)"}
        if ctx.scalar_reduce("ipc") == "ipc":
            try:
                dt_ipc, stats_ipc = timed_iterations()
                failed = 0.0
            except Exception as e:                  # e.g. a peer's stamp never arrived (KRYST_ERR_RCCL after the poll budget)
                dt_ipc, stats_ipc, failed = float("inf"), None, 1.0
                reduce_info["ipc"] = f"failed: {e}"
            if group.allreduce_max(failed) > 0.0:    # one rank's failure is everybody's: back to the all-gather path, on every rank
                reduce_info.setdefault("ipc", "failed on another rank")
                if not reduce_info["ipc"].startswith("failed"):
                    reduce_info["ipc"] = "failed on another rank"
                ctx.scalar_reduce("rccl")
            else:
                same = group.allreduce_max(0.0 if stats_ipc.final_residual == stats.final_residual else 1.0) == 0.0
                reduce_info.update(value_ipc=steps / dt_ipc, ipc="bit-identical residual" if same else "DIFFERENT residual: not used")
                if same and dt_ipc < dt:
                    dt, stats = dt_ipc, stats_ipc
                    reduce_info["path"] = "ipc"
                else:
                    ctx.scalar_reduce("rccl")
    # N > 1: the halo exchange of the new p started behind the direction pass's boundary tiles (the default) against started by the next
    # SpMV (KRYST_HALO_EARLY=0): the same bits; the faster one is the headline, both figures are reported
    if world > 1:
        with env_override(KRYST_HALO_EARLY="0"):
            dt_late, stats_late = timed_iterations()
        same = group.allreduce_max(0.0 if stats_late.final_residual == stats.final_residual else 1.0) == 0.0
        reduce_info.update(value_halo_early=steps / dt, value_halo_at_spmv=steps / dt_late, halo="early")
        if same and dt_late < dt:
            dt, stats = dt_late, stats_late
            reduce_info["halo"] = "at the SpMV"
            os.environ["KRYST_HALO_EARLY"] = "0"          # (the plain-CSR and phase runs below use the same setting)
    enc = a.encoding()
    with env_override(KRYST_SPMV_COMPRESS="0"):
        dt_plain, stats_plain = timed_iterations()
    # the dominant kernel, timed live with HIP events on the compute stream (kryst_bench_spmv), in both storage forms
    y = ctx.vec(nloc)
    # (three averages of 20 / 10 back-to-back launches each, the median of the three: one disturbed batch -- seen once, a 7x outlier --
    # must not become the roofline figure)
    spmv_ms = sorted(a.bench_spmv(b, y, fused_dots=1, reps=20) for _ in range(3))[1]
    with env_override(KRYST_SPMV_COMPRESS="0"):
        plain_ms = sorted(a.bench_spmv(b, y, fused_dots=1, reps=10) for _ in range(3))[1]
    # context: the device-copy rate at this footprint (hipMemcpy D2D of one vector, read + write)
    y.copy_from(b); ctx.synchronize()
    ctx.timer_start()
    for _ in range(10):
        y.copy_from(b)
    copy_gbs = 10 * 16.0 * nloc / (ctx.timer_stop() * 1e-3) / 1e9
    # where an iteration's device time goes (a separate short run with event marks between the phases)
    phases = None
    if phase_iters > 0:
        x = ctx.vec(nloc)
        sess = K.Session(solver, a, pc, b, x, tol=0.0, max_iters=phase_iters + 2)
        sess.step(2)
        barrier()
        ctx.phase_timing_begin()
        sess.step(phase_iters)
        ph = ctx.phase_timing_end()
        sess.end()
        mine = {k: v / phase_iters for k, v in ph.items() if v > 0.0}
        mine["total"] = sum(mine.values())
        phases = group.gather(mine)
    # N > 1: what one scalar collective costs end to end (local value -> RCCL all-gather -> rank-ordered fold -> host), so that the
    # first real multi-GPU run says how much of an iteration the two inner-product exchanges can be at most
    collective_us = None
    if world > 1:
        for _ in range(10):
            ctx.all_reduce(1.0)
        barrier()
        t0 = time.perf_counter()
        for _ in range(100):
            ctx.all_reduce(1.0)
        collective_us = group.allreduce_max((time.perf_counter() - t0) / 100 * 1e6)
    nnz_loc = a.nnz
    return {"dt": dt, "dt_plain": dt_plain, "stats": stats, "enc": enc, "nloc": nloc, "nnz_loc": nnz_loc, "collective_us": collective_us, "reduce_info": reduce_info,
            "roofline": roofline_of(enc, grid, nloc, nnz_loc, spmv_ms, world, staged=a.pattern_info()["staged"]),
            "roofline_csr": roofline_csr_of(grid, nloc, nnz_loc, plain_ms, world),
            "blas1": blas1_streams(K, ctx, nloc), "copy_gbs": copy_gbs, "phases": phases,
            "final_residual_plain": stats_plain.final_residual}


def timed_solve(K, ctx, make_solver, a, pc, b, repeat=2):
    """Device-resident solve to the solver's tolerance, timed on the host around the call (device idle before and after);
    solved `repeat` times, the faster one reported (the first grows the context's work arena)."""
    best = None
    for _ in range(repeat):
        s = make_solver()
        x = ctx.vec(a.nrows())
        ctx.synchronize(); t0 = time.perf_counter()
        st = s.solve(a, pc, b, x)
        ctx.synchronize(); dt = time.perf_counter() - t0
        if best is None or dt < best[0]:
            best = (dt, st)
    return best


def stream_block(ctx, n, kind, name, words, reps=20):
    import ctypes as C
    from kryst_amd._ffi import lib, check
    stride = ((n + 511) // 512 * 512 + 512) * 8
    ms = C.c_double(0)
    check(lib().kryst_bench_streams(ctx.h, n, stride, kind, reps, C.byref(ms)))
    ach = words * 8 * n / (ms.value * 1e-3) / 1e9
    return {"bound": "hbm", "kernel": name, "bytes_per_launch": words * 8 * n, "ms_per_launch": ms.value, "achieved": ach,
            "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": None}


def tri_roofline(pc, r, z, n, nnz, reps=20):
    """The ILU apply (forward + backward triangular solve), timed live with HIP events.  `frac` prices it at the bytes the kernels
    MOVE by their own bookkeeping (coefficient chunks actually requested + right-hand side read + result written, both
    directions); SURVEY 8(d)'s B_spmv + 8 n is reported beside it as `algorithmic_*` and is NOT a fraction of moved bytes when
    repeating coefficient chunks are skipped."""
    ms = pc.bench_apply(r, z, reps)
    info = pc.ilu_info()
    alg = spmv_bytes(n, nnz) + 8 * n
    out = {"bound": "hbm", "kernel": "ILU(0) apply = forward + backward solve, " + info["form"], "ms_per_apply": ms, "peak": HBM_PEAK_GBS, "unit": "GB/s",
           "algorithmic_bytes": alg, "algorithmic_model": "SURVEY 8(d): B_spmv + 8 n", "algorithmic_GBs": alg / (ms * 1e-3) / 1e9,
           "algorithmic_frac_note": "algorithmic_GBs / peak is not a roofline fraction where coefficient chunks are skipped", "traffic": None, "form": info}
    if info["form"].startswith("grid 16x16"):
        ch, sk, bpc = info["chunks"], info["chunks_not_requested"], info["bytes_per_chunk"]
        moved = (ch[0] - sk[0]) * bpc[0] + (ch[1] - sk[1]) * bpc[1] + 32 * n
        out.update(bytes_moved=moved, bytes_model="coefficient chunks requested x bytes per chunk + 32 n (r, y read; y, z written)",
                   achieved=moved / (ms * 1e-3) / 1e9, frac=moved / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                   chunks_not_requested_frac=(sk[0] + sk[1]) / max(1, ch[0] + ch[1]))
    else:
        out.update(bytes_moved=None, achieved=None, frac=None)
    return out


def other_configs(K, ctx, steps, warmup):
    """BASELINE configs 3, 5 (256^3, solved to their tolerance), config 4's workload on one GPU (512^3 Jacobi-PCG, stepped) and the
    variable-coefficient operator at 256^3 / 512^3.  N = 1 only."""
    out = {}
    # ---- config 3: GMRES(30) Left + Jacobi on 256^3 convection-diffusion, tol 1e-8, max 600 (gmres.rs:216-402)
    a = K.CsrMatrix.stencil7(256, "convdiff", ctx=ctx); n = a.nrows()
    b = a.spmv(ctx.vec(n).fill(1.0))
    pc = K.Jacobi().setup(a)
    dt, st = timed_solve(K, ctx, lambda: K.GmresSolver(30, 1e-8, 600), a, pc, b)
    out["config3_gmres30_jacobi_256"] = {
        "workload": "gmres30_left_jacobi_convdiff7_256^3", "value": st.iterations / dt, "unit": "iterations/s", "iterations": st.iterations,
        "converged": bool(st.converged), "final_residual": st.final_residual, "solve_seconds": dt, "spmv_encoding": a.encoding()[0],
        "algorithmic_bytes_per_iteration": spmv_bytes(n, a.nnz) + 1288 * n,
        "roofline": stream_block(ctx, n, 0, "ew_kernel<MgsLinkOp> (Gram-Schmidt link z -= h v_i fused with the next link's dot: 3 reads + 1 write; "
                                 "31 links per iteration on average)", 4)}
    del pc, a, b
    # ---- config 5: right-preconditioned BiCGStab + true ILU(0) on 256^3 anisotropic Poisson, absolute tol 1e-8 ||b|| (bicgstab.rs:69-293)
    a = K.CsrMatrix.stencil7(256, "aniso", ctx=ctx); n = a.nrows()
    b = a.spmv(ctx.vec(n).fill(1.0)); bn = K.norm(b)
    ctx.synchronize(); t0 = time.perf_counter()
    pc = K.TrueIlu0().setup(a)
    ctx.synchronize(); t_setup = time.perf_counter() - t0
    dt, st = timed_solve(K, ctx, lambda: K.BiCgStabRightPcSolver(1e-8 * bn, 3000), a, pc, b)
    z = ctx.vec(n)
    out["config5_bicgstab_ilu0_256"] = {
        "workload": "bicgstab_right_true_ilu0_aniso7_256^3", "value": st.iterations / dt, "unit": "iterations/s", "iterations": st.iterations,
        "converged": bool(st.converged), "final_residual": st.final_residual, "solve_seconds": dt, "ilu_setup_ms": t_setup * 1e3,
        "spmv_encoding": a.encoding()[0], "note": "the reference's BiCGStab ignores pc (bicgstab.rs:70); the preconditioned form is a labelled extension",
        "roofline": tri_roofline(pc, b, z, n, a.nnz)}
    del pc, a, b, z
    # ---- config 4's workload on ONE GPU: Jacobi-PCG on 512^3 Poisson (pcg.rs:114-222), fixed stepped iterations
    k4 = min(steps, 100)
    a = K.CsrMatrix.stencil7(512, "poisson", ctx=ctx); n = a.nrows()
    b = a.spmv(ctx.vec(n).fill(1.0))
    pc = K.Jacobi().setup(a)
    x = ctx.vec(n)
    with K.Session("pcg", a, pc, b, x, tol=0.0, max_iters=warmup + k4) as sess:
        sess.step(warmup); ctx.synchronize(); t0 = time.perf_counter()
        sess.step(k4); ctx.synchronize(); dt = time.perf_counter() - t0
        st = sess.end()
    out["config4_jacobi_pcg_512"] = {
        "workload": "jacobi_pcg_poisson7_512^3 on one GPU (the 8-way partition is the --gpus 8 run)", "value": k4 / dt, "unit": "iterations/s", "steps": k4,
        "ms_per_step": dt / k4 * 1e3, "final_residual": st.final_residual, "spmv_encoding": a.encoding()[0],
        "algorithmic_bytes_per_iteration": spmv_bytes(n, a.nnz) + 136 * n}
    del pc, a, b, x
    # ---- variable coefficients: no row patterns, no value dictionary, no repeating coefficient chunks
    for grid in (256, 512):
        kv = min(steps, 100)
        a = K.CsrMatrix.stencil7(grid, "varcoef", ctx=ctx); n = a.nrows(); nnz = a.nnz
        b = a.spmv(ctx.vec(n).fill(1.0)); y = ctx.vec(n)
        enc = a.encoding()
        blk = {"workload": f"variable-coefficient 7-point diffusion operator, {grid}^3 (kind varcoef: per-edge weights from splitmix64)", "spmv_encoding": enc[0]}
        for form, env in (("default", {}), ("plain_csr", {"KRYST_SPMV_COMPRESS": "0"})):
            with env_override(**env):
                x = ctx.vec(n)
                with K.Session("cg", a, None, b, x, tol=0.0, max_iters=warmup + kv) as sess:
                    sess.step(warmup); ctx.synchronize(); t0 = time.perf_counter()
                    sess.step(kv); ctx.synchronize(); dt = time.perf_counter() - t0
                    sess.end()
                ms = a.bench_spmv(b, y, fused_dots=1, reps=20)
                e = a.encoding()
            if form == "default":
                blk.update(value=kv / dt, unit="iterations/s", steps=kv, ms_per_step=dt / kv * 1e3, roofline=roofline_of(e, grid, n, nnz, ms, 1, "varcoef"))
            else:
                blk.update(value_plain_csr=kv / dt, ms_per_step_plain_csr=dt / kv * 1e3, roofline_csr=roofline_csr_of(grid, n, nnz, ms, 2))
            del x
        ctx.synchronize(); t0 = time.perf_counter()
        pc = K.TrueIlu0().setup(a)
        ctx.synchronize(); blk["ilu_setup_ms"] = (time.perf_counter() - t0) * 1e3
        blk["ilu_apply"] = tri_roofline(pc, b, y, n, nnz, reps=10)
        out[f"variable_coefficient_{grid}"] = blk
        del pc, a, b, y
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--grid", type=int, default=0, help="grid edge (default 512 for every N)")
    ap.add_argument("--solver", default="cg", choices=["cg", "pcg"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-256", action="store_true", help="skip the config1_256 block (N = 1 measures BASELINE configs[1]'s 256^3 grid too)")
    ap.add_argument("--no-configs", action="store_true", help="skip the other BASELINE configs and the variable-coefficient blocks (N = 1)")
    ap.add_argument("--phase-iters", type=int, default=20, help="iterations of the per-phase timing run (0: skip)")
    ap.add_argument("--launcher", default="auto", choices=["auto", "torch", "socket"],
                    help="N > 1 plumbing for the RCCL id / barrier: torch.distributed gloo (default) or kryst_amd/launch.py (no torch)")
    args = ap.parse_args()

    # a hung collective must not hang the node: give up loudly after 20 minutes
    def _watchdog():
        sys.stderr.write("bench.py: watchdog timeout (1200 s), aborting\n"); sys.stderr.flush(); os._exit(124)
    wd = threading.Timer(1200.0, _watchdog); wd.daemon = True; wd.start()

    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: what RCCL needs between processes on this driver
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N with N > 1 must be launched by torch.distributed.run (one rank per GPU)")
        args.gpus = world
    grid = args.grid or 512          # ONE fixed problem for every N (strong scaling, north_star): 512^3

    group = make_group(args.launcher, rank, world)
    import kryst_amd as K
    if world > 1:
        uid = group.broadcast_bytes(K.Context.unique_id() if rank == 0 else None)
        # KRYST_BENCH_DEVICE: rehearsal of the N > 1 flow on a one-GPU box (all ranks on one device, RCCL stand-in)
        dev = int(os.environ.get("KRYST_BENCH_DEVICE", local_rank))
        ctx = K.Context(dev, rank, world, uid)
    else:
        ctx = K.Context(0)

    n = grid ** 3
    nnz = 7 * n - 6 * grid * grid
    m = measure(K, ctx, group, grid, args.solver, args.warmup, args.steps, args.phase_iters)
    wl = "jacobi_pcg" if args.solver == "pcg" else "cg"
    out = {
        "metric": "cg_iterations_per_sec", "value": args.steps / m["dt"], "unit": "iterations/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": m["dt"] / args.steps * 1e3,
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"{wl}_poisson7_{grid}^3", "grid": grid, "rows": n, "nnz": nnz, "partition": f"{world} k-slab(s)",
                   "rhs": "A*1", "x0": "0", "final_residual": m["stats"].final_residual, "spmv_encoding": m["enc"][0],
                   "launcher": type(group).__name__},
        "value_plain_csr": args.steps / m["dt_plain"], "ms_per_step_plain_csr": m["dt_plain"] / args.steps * 1e3,
        "roofline": m["roofline"], "roofline_csr": m["roofline_csr"], "roofline_blas1": m["blas1"],
        "measured_copy_GBs": m["copy_gbs"],
        "phase_ms": m["phases"],         # per rank: device ms per iteration by phase (spmv / halo_wait / spmv_boundary / reduce / blas1)
        "scalar_all_reduce_us": m["collective_us"],   # N > 1: host round trip of one scalar all-reduce (all-gather + ordered fold + sync), max over ranks
        "scalar_reduce": m["reduce_info"],            # N > 1: which path carried the inner products in `value` (RCCL all-gather or hipIpc mailboxes) and both figures
    }
    base256 = None
    if world == 1 and grid != 256 and not args.no_256:
        m2 = measure(K, ctx, group, 256, args.solver, args.warmup, args.steps, args.phase_iters)
        out["config1_256"] = {"workload": f"{wl}_poisson7_256^3", "value": args.steps / m2["dt"], "unit": "iterations/s",
                              "ms_per_step": m2["dt"] / args.steps * 1e3, "value_plain_csr": args.steps / m2["dt_plain"],
                              "roofline": m2["roofline"], "roofline_csr": m2["roofline_csr"], "roofline_blas1": m2["blas1"],
                              "phase_ms": m2["phases"]}
        if not args.no_cpu_baseline:
            try:
                base256 = cpu_cg(256, 8.0)
                out["config1_256"]["cpu_baseline"] = base256
            except Exception as e:
                out["config1_256"]["cpu_baseline"] = {"value": None, "unit": "cg_iterations/s", "cores": 0, "kind": "port", "sample": f"failed: {e}"}
    if world == 1 and not args.no_configs:
        try:
            out.update(other_configs(K, ctx, args.steps, args.warmup))
        except Exception as e:                              # the headline line must survive a failing side measurement
            out["other_configs_error"] = f"{type(e).__name__}: {e}"
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(grid, base256)
    if rank == 0:
        print(json.dumps(out), flush=True)
    group.close()


if __name__ == "__main__":
    main()
