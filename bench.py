#!/usr/bin/env python3
"""bench.py -- CG iterations/sec + SpMV GB/s against the HBM roofline on synthetic 3-D 7-point Poisson CSR.

    python bench.py --gpus N --steps K --warmup W
        N > 1 started PLAINLY (no WORLD_SIZE in the environment): this process launches N rank processes of itself, one per GPU (self_launch below);
        under `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...` (RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* set) it is one rank.

A "step" is ONE iteration of the reference's unpreconditioned CG (src/solver/cg.rs:141-285: SpMV + 2 inner products + 3 vector updates) on a matrix,
right-hand side and iterate that are already resident in HBM.  W warm-up iterations and three batches of exactly K timed iterations run inside one
stepping session (tol = 0, so the device never stops early); every batch is bracketed by device synchronize + barrier on both sides, the maximum over
ranks is taken per batch and the MEAN of the batches is the headline (their median when one batch is disturbed: slowest > 1.1 x fastest).  One JSON line on rank 0.  (--solver pcg / gmres: Jacobi-PCG / GMRES(30) + Jacobi.)

Workload: ONE fixed 512^3 system for every N (BASELINE.json's metric is quoted on 256^3 / 512^3; north_star asks for strong scaling of CG iterations/sec
on 512^3, which fits a single 288 GB GPU).  N > 1: row-partitioned in k-slabs over the ranks, halo planes and inner products over xGMI (hipIpc peer
stores / mailboxes where a checked test transfer succeeds on every rank, else RCCL; the line says which).

What the line holds (every `frac` is bytes the named kernel really moves / its HIP-event time / 8 TB/s, so it is <= 1):
  value             CG iterations/s with the operator in its default (most compact lossless) storage form
  value_sec8d       the same K iterations with KRYST_SPMV_COMPRESS=0 (= value_plain_csr): the 12-bytes-per-entry CSR arrays of SURVEY 8(d) are streamed,
                    which is what every matrix that is not a constant-coefficient stencil gets -- the figure comparable to north_star
  roofline          the DOMINANT kernel of the timed loop, priced at the bytes IT streams, timed inside the solver's iterations, PMC `traffic` of exactly that
                    kernel beside it (at N = 1 measured by this very command: two rocprofv3 --pmc passes in child processes).  At 512^3 on one GPU that is
                    spmv_pattern_fuse_kernel (direction pass + SpMV + (p,Ap) partials in one launch, x updated in batches: 34 bytes per row); otherwise the
                    SpMV kernel of the default storage form.  Inside it:
                      roofline.sec8d                 the plain-CSR kernel on SURVEY 8(d)'s bytes (north_star's "% of HBM roofline on CSR SpMV"), its traffic
                                                     skeleton on the same arrays, the homes tried for the arrays, value_sec8d
                      roofline.x_batch               the pass that pays x += alpha_i p_i for a batch of iterations
                      roofline.spmv_alone            (fused loop only) the staged-window SpMV kernel of the default form on its own, as earlier rounds reported it
                      roofline.fused_direction_spmv  (fused loop only) the top-level figures again under the key this round introduced first
  roofline_blas1    the vector kernels of a CG iteration, timed INSIDE the solver's iterations (phase run)
  gmres30_jacobi    GMRES(30) Left + Jacobi on the same operator and partition: iterations/s of solves of exactly 60 iterations (north_star: CG / GMRES at 1-8 GPUs)
  phase_ms          device time per iteration by phase (hipEvents between the phases, a separate short run), per rank
  scalar_reduce     (N > 1) which transport carried the inner products and the halo in `value`, every form's figure, the library's defaults
  config1_256       (N = 1) the same measurements on BASELINE configs[1]'s 256^3 grid, with the CPU port timed on that grid
  config3_gmres30_jacobi_256 (three GMRES forms with their true residuals), config5_bicgstab_ilu0_256, config4_jacobi_pcg_512, variable_coefficient_256 / _512,
  general_ilu       (N = 1) the other BASELINE configs and operators the encodings do not apply to, each with the roofline block of ITS dominant kernel
  cpu_baseline      the oracle's CG timed on the host cores AT the workload's size (16 threads and all allowed CPUs, the CPU named); parity_at_size: its
                    iterations compared with the GPU bit for bit
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

    env_leader = True

    def __init__(self, rank, world):
        import torch.distributed as dist
        self.dist, self.rank, self.world = dist, rank, world
        # gloo announces its connections on STDOUT ("[Gloo] Rank 0 is connected to ..."), from C++: the line the driver reads must be the only
        # thing on rank 0's stdout, so file descriptor 1 points at stderr while the group is set up and warmed
        sys.stdout.flush()
        saved = os.dup(1)
        try:
            os.dup2(2, 1)
            dist.init_process_group("gloo", rank=rank, world_size=world)
            dist.barrier()
        finally:
            sys.stdout.flush()
            os.dup2(saved, 1)
            os.close(saved)

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
    rank, world, env_leader = 0, 1, True
    def broadcast_bytes(self, payload): return payload
    def barrier(self): pass
    def allreduce_max(self, x): return x
    def gather(self, obj): return [obj]
    def close(self): pass


class ThreadedGroup:
    """Several ranks of ONE process, a host thread each (rehearsals on a one-GPU box, which admits at most 6 processes on its card: 8 ranks =
    4 processes x 2 threads).  Every method is collective over ALL ranks: the threads of a process meet at a barrier, thread 0 talks to the
    other processes through the inner group."""

    class Shared:
        def __init__(self, inner, per):
            self.inner, self.per = inner, per
            self.bar = threading.Barrier(per)
            self.slots = [None] * per
            self.result = None

    def __init__(self, shared, t):
        self.sh, self.t = shared, t
        self.rank = shared.inner.rank * shared.per + t
        self.world = shared.inner.world * shared.per
        self.env_leader = t == 0

    def _via_leader(self, value, combine_local, inner_call):
        sh = self.sh
        sh.slots[self.t] = value
        sh.bar.wait()
        if self.t == 0:
            sh.result = inner_call(combine_local(list(sh.slots)))
        sh.bar.wait()
        out = sh.result
        sh.bar.wait()
        return out

    def broadcast_bytes(self, payload):
        return self._via_leader(payload, lambda v: v[0], self.sh.inner.broadcast_bytes)

    def barrier(self):
        self._via_leader(0, lambda v: 0, lambda _: self.sh.inner.barrier())

    def allreduce_max(self, x):
        return self._via_leader(float(x), max, self.sh.inner.allreduce_max)

    def gather(self, obj):
        return self._via_leader(obj, lambda v: v, lambda v: [o for part in self.sh.inner.gather(v) for o in part])

    def close(self):
        self._via_leader(0, lambda v: 0, lambda _: self.sh.inner.close())


_T0 = time.time()
_STAGE = {}
_PROVISIONAL = {}                  # "line": a complete, measured JSON line the watchdog prints (rank 0) when a LATER stage hangs


def stage(rank, name):
    """Flushed stage marker on stderr: when a run hangs or dies, the tail the driver keeps names the phase."""
    _STAGE[rank] = name
    sys.stderr.write(f"[bench rank {rank} +{time.time() - _T0:7.2f}s] {name}\n")
    sys.stderr.flush()


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
        g = Rendezvous.from_env()
        g.env_leader = True
        return g
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


def cpu_info():
    """CPU model, sockets and cores of this host (/proc/cpuinfo) and the CPUs this process may run on ("CPU model,
    socket/core count"; the reference sizes its Rayon pool with num_cpus::get(), src/parallel/rayon_comm.rs:29-35)."""
    model, phys, cores, logical = None, set(), set(), 0
    try:
        pid = cid = None
        for line in open("/proc/cpuinfo"):
            k, _, v = line.partition(":")
            k, v = k.strip(), v.strip()
            if k == "processor":
                logical += 1
            elif k == "model name" and model is None:
                model = v
            elif k == "physical id":
                pid = v; phys.add(v)
            elif k == "core id":
                cid = v; cores.add((pid, cid))
    except Exception:
        pass
    # the cgroup's CPU bandwidth (cpu.max = "quota period"): a container may be ALLOWED on every CPU of the host and still be granted only a
    # few CPUs' worth of time -- threads beyond that are throttled, not run (the GPU boxes: 256 CPUs allowed, 16 granted)
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = float(q) / float(per)
    except Exception:
        pass
    return {"cpu_model": model, "sockets": len(phys) or None, "physical_cores": len(cores) or None, "logical_cpus": logical or None,
            "cpus_allowed": len(os.sched_getaffinity(0)), "cpu_quota_cores": quota}


def cpu_cg(grid, seconds, ctx=None):
    """The oracle's CG (the CPU restatement of the reference path, OpenMP over rows / tiles like the reference's Rayon loops,
    device-order dot) on the grid^3 Poisson system: as many iterations as fit in ~`seconds`, on ALL the CPUs this process may use
    (num_cpus::get() threads, rayon_comm.rs:29-35) and -- when that is more than 16 -- on 16 threads (the GPU box's CPU share for one
    GPU) as well; `value` is the faster of the two and `cores` its thread count.  -> dict.
    With a device context the oracle's residual history -- which this leg computes anyway -- CHECKS the GPU at full size: the same
    system solved by the library for the same number of iterations, histories compared entry by entry (`parity_at_size`; the oracle is
    the checker, outside every timed region)."""
    import numpy as np
    import kryst_amd as K
    from oracle import oracle as O
    info = cpu_info()
    # "all the CPUs this process may use": the affinity mask, cut to the cgroup's CPU bandwidth when that is smaller (256 threads on a quota of 16
    # CPUs are throttled to 16 CPUs' worth of time and thrash: round 5 measured 0.49 it/s that way against 4.4 on 16 threads)
    cores_all = info["cpus_allowed"]
    if info.get("cpu_quota_cores"):
        cores_all = max(1, min(cores_all, int(info["cpu_quota_cores"] + 0.999)))
    T, V, F = K.reduce_spec()
    rp, ci, va = K.host_stencil7(grid, "poisson")
    a = O.Csr(grid ** 3, grid ** 3, rp, ci, va, check=False)
    b = a.spmv(np.ones(a.nrows))
    rs = O.Reduce.tiled(T, V, F)

    def run(threads, budget):
        O.set_threads(threads)
        t0 = time.perf_counter(); O.solve("cg", a, b, tol=0.0, max_iters=2, rs=rs); t2 = time.perf_counter() - t0
        iters = int(max(3, min(400, budget / max(t2 / 3.0, 1e-4))))     # 2 iterations + the initial residual pass
        t0 = time.perf_counter()
        res = O.solve("cg", a, b, tol=0.0, max_iters=iters, rs=rs)
        return res, time.perf_counter() - t0

    # 16 threads first (the figure every earlier round quoted, and on a two-socket host usually the faster one: the arrays are first touched by
    # one thread); its iterations are the ones the GPU is checked against below.  Then every allowed CPU (num_cpus::get(), what BASELINE.md promises).
    t16 = min(16, cores_all)
    res, dt = run(t16, seconds)
    out = {"unit": "cg_iterations/s", "kind": "port", "grid": grid, "extrapolated": False, "cores_all": cores_all}
    out.update(info)
    out["value_16_threads"] = res.iterations / dt
    sample = (f"{res.iterations} oracle CG iterations on the {grid}^3 Poisson system in {dt:.1f} s (OpenMP rows/tiles over {t16} threads, device-order dot; "
              "usize = int64 indices like the reference)")
    if cores_all > t16:
        r_all, dt_all = run(cores_all, seconds / 2)
        out["value_all_cores"] = r_all.iterations / dt_all
        sample += f"; {r_all.iterations} iterations in {dt_all:.1f} s on all {cores_all} usable CPUs"
    else:
        out["value_all_cores"] = out["value_16_threads"]
        if info.get("cpu_quota_cores") and info["cpus_allowed"] > cores_all:
            sample += (f"; the process is allowed on {info['cpus_allowed']} CPUs but its cgroup grants {info['cpu_quota_cores']:g} CPUs of time (cpu.max): "
                       f"{cores_all} threads ARE all the cores it can use")
    best_all = out["value_all_cores"] > out["value_16_threads"]
    out["value"], out["cores"] = (out["value_all_cores"], cores_all) if best_all else (out["value_16_threads"], t16)
    out["sample"] = sample + f"; `value` = the faster figure ({out['cores']} threads)"
    O.set_threads(t16)
    iters = res.iterations
    if ctx is not None:
        try:
            del a, rp, ci, va
            ga = K.CsrMatrix.stencil7(grid, "poisson", ctx=ctx)
            gb = ga.spmv(ctx.vec(ga.nrows()).fill(1.0))
            same_b = bool(np.array_equal(gb.to_host(), b))
            gs = K.CgSolver(0.0, iters)
            gx = ctx.vec(ga.nrows())
            gs.solve(ga, None, gb, gx)
            gh, oh = np.array(gs.residual_history), np.array(res.history)
            m = min(len(gh), len(oh))
            dev = float(np.max(np.abs(gh[:m] - oh[:m])) / oh[0]) if m else None
            out["parity_at_size"] = {"grid": grid, "solver": "cg (cg.rs:141-288), tol 0", "iterations_compared": int(m - 1), "history_entries": int(m),
                                     "rhs_bit_identical": same_b, "bit_identical": bool(len(gh) == len(oh) and np.array_equal(gh, oh)),
                                     "max_rel_dev": dev, "x_bit_identical": bool(np.array_equal(gx.to_host(), res.x)),
                                     "checker": "oracle/kryst_oracle.c kro_cg in the library's dot order (KRO_REDUCE_TILED), " + str(t16) + " threads"}
            del ga, gb, gx
        except Exception as e:
            out["parity_at_size"] = {"grid": grid, "error": f"{type(e).__name__}: {e}"}
    if grid <= 256:                                     # the bit-canonical single-thread figure beside it, a few iterations
        rp, ci, va = K.host_stencil7(grid, "poisson")
        a = O.Csr(grid ** 3, grid ** 3, rp, ci, va, check=False)
        O.set_threads(1)
        t0 = time.perf_counter(); r1 = O.solve("cg", a, b, tol=0.0, max_iters=4, rs=rs); dt1 = time.perf_counter() - t0
        O.set_threads(t16)
        out["single_thread_value"] = r1.iterations / dt1
    return out


def cpu_baseline(grid, base256, ctx=None):
    """Measured AT the workload's size when the host has the memory for it (512^3: 24 GB of int64-indexed CSR + vectors),
    otherwise the 256^3 measurement scaled by the row count and marked as extrapolated."""
    try:
        if grid <= 256:
            return base256 or cpu_cg(grid, 8.0, ctx)
        need = 16.0 * 7 * grid ** 3 / 1e9 + 8 * 8.0 * grid ** 3 / 1e9 + 6.0
        if os.environ.get("KRYST_BENCH_CPU_FULL", "1") != "0" and host_mem_available_gb() > need:
            return cpu_cg(grid, 10.0, ctx)
        b = base256 or cpu_cg(256, 8.0, ctx)
        scale = (256 / grid) ** 3
        out = dict(b, value=b["value"] * scale, grid=grid, extrapolated=True)
        out["sample"] = b["sample"] + f"; scaled by {scale:.4f} = (256/{grid})^3 rows (host memory below {need:.0f} GB)"
        for k in ("single_thread_value", "value_all_cores", "value_16_threads"):
            if k in out:
                out[k] = b[k] * scale
        out.pop("parity_at_size", None)
        return out
    except Exception as e:                              # the oracle is only the reported baseline, never the product
        return {"value": None, "unit": "cg_iterations/s", "cores": 0, "kind": "port", "sample": f"failed: {e}"}


# ---------------------------------------------------------------------------------------------------------------- GPU side
_LIVE_TRAFFIC = {}
LIVE_FORMS = {(512, "default"), (512, "plain"), (512, "fused")}     # measured in THIS run (about 7 s per pass); the other sizes / forms come from profiles/


def live_traffic(grid, form):
    """HBM bytes per SpMV launch measured in THIS run: two rocprofv3 --pmc passes (FETCH_SIZE, then WRITE_SIZE -- separate runs, no
    trace domains, as MI355X_MICROARCH.md prescribes) over tools/spmv_only.py in child processes; the read side is calibrated on the
    ew_kernel<DotOp> launches of the same process, whose byte count is known (the counter under-reports wide coalesced reads on
    gfx950).  None when rocprofv3 is missing, a pass fails or KRYST_BENCH_LIVE_TRAFFIC=0."""
    key = (grid, form)
    if key in _LIVE_TRAFFIC:
        return _LIVE_TRAFFIC[key]
    res = None
    # (never under an outer profiler: the children would inherit its preloaded tool library and the inner rocprofv3 would exec with the
    # GPU already initialised -- ADVICE r03; tools/profile_round.sh also switches the live passes off)
    profiled = any(k == "LD_PRELOAD" and "rocprof" in v or k.startswith(("ROCP_", "ROCPROF")) or k == "HSA_TOOLS_LIB" for k, v in os.environ.items())
    if os.environ.get("KRYST_BENCH_LIVE_TRAFFIC", "1") != "0" and key in LIVE_FORMS and not profiled:
        import collections, csv, glob, shutil, subprocess, tempfile
        exe = shutil.which("rocprofv3")
        tmp = tempfile.mkdtemp(prefix="kryst_pmc_", dir="/tmp") if exe else None
        try:
            env = {k: v for k, v in os.environ.items() if k not in ("LD_PRELOAD", "HSA_TOOLS_LIB") and not k.startswith(("ROCP_", "ROCPROF"))}
            env["TMPDIR"] = "/tmp"
            env.pop("KRYST_SPMV_COMPRESS", None)
            if form == "plain":
                env["KRYST_SPMV_COMPRESS"] = "0"
            means = {}
            # form "fused": the kernel the timed CG loop launches when the direction pass rides inside the SpMV -- a CG session of 26 iterations
            # (tools/cg_only.py; it ends with the DotOp launches the read side is calibrated on, like spmv_only.py)
            prog = ([os.path.join(ROOT, "tools", "cg_only.py"), str(grid), "26"] if form == "fused" else
                    [os.path.join(ROOT, "tools", "spmv_only.py"), str(grid), "5", "1", "varcoef" if form == "varcoef" else "poisson"])
            for counter in ("FETCH_SIZE", "WRITE_SIZE"):
                d = os.path.join(tmp, counter)
                r = subprocess.run([exe, "--pmc", counter, "--output-format", "csv", "-d", d, "-o", "p", "--", sys.executable] + prog,
                                   cwd="/tmp", env=env, capture_output=True, timeout=90)
                if r.returncode != 0:
                    raise RuntimeError(f"rocprofv3 --pmc {counter} failed")
                acc = collections.defaultdict(list)
                for row in csv.DictReader(open(glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True)[0])):
                    if row["Counter_Name"] == counter:
                        acc[row["Kernel_Name"]].append(float(row["Counter_Value"]))
                means[counter] = {k: sum(v) / len(v) for k, v in acc.items()}
            spmv = [k for k in means["FETCH_SIZE"] if ("spmv_pattern_fuse_kernel" in k if form == "fused" else "spmv" in k and "<1," in k)][0]
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
           "frac_of_streamed_bytes": ach / HBM_PEAK_GBS,
           "frac_basis": "`frac` prices the launch at the bytes THIS storage form streams; SURVEY 8(d)'s plain-CSR bytes / time / peak is "
                         "`frac_of_sec8d_bytes` (it exceeds 1 when the form streams fewer bytes than CSR does -- it is NOT a bandwidth fraction); the "
                         "roofline fraction of the kernel that moves SURVEY 8(d)'s bytes is `roofline_csr.frac`",
           "frac_of_sec8d_bytes": alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
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


def blas1_in_solver(K, ctx, n, solver, phase):
    """roofline_blas1 from the SOLVER'S OWN kernels: the hipEvent time the phase run charged to the residual pass and to the direction pass of
    this rank's iterations (kryst_phase_timing: "blas1_residual" / "blas1_direction"), priced at the words those kernels move.  The same kernel
    shapes timed in isolation on fresh vectors (kryst_bench_streams) are kept beside them as `isolated_ms_per_launch` -- round 4 quoted only
    those, and they ran 13 % slower than the kernels inside the solve."""
    iso = blas1_streams(K, ctx, n) if solver in ("cg", "pcg") else []
    if not phase or solver not in ("cg", "pcg") or "blas1_residual" not in phase:
        return iso
    # (no "blas1_direction" phase: the direction pass is inside the SpMV -- spmv_pattern_fuse_kernel -- and only the residual pass is a BLAS-1 launch)
    defer = os.environ.get("KRYST_CG_DEFER_X", "1") != "0"
    if solver == "cg":
        shapes = (("blas1_residual", "ew_kernel<CgResidualOp> (r -= alpha Ap, fused (r,r))", 3), ("blas1_direction", "ew_kernel<CgDirectionOp> (x += alpha p, p = r + beta p)", 5)) if defer else \
                 (("blas1_residual", "ew_kernel<CgUpdate1> (x += alpha p, r -= alpha Ap, fused (r,r))", 6), ("blas1_direction", "ew_kernel<AypxDevOp> (p = r + beta p)", 3))
    else:
        shapes = (("blas1_residual", "ew_kernel<PcgResidualOp> (r -= alpha Ap, z = D^-1 r, fused (r,z), (r,r))", 5), ("blas1_direction", "ew_kernel<CgDirectionOp> (x += alpha p, p = z + beta p)", 5)) if defer else \
                 (("blas1_residual", "ew_kernel<PcgUpdateOp> (x += alpha p, r -= alpha Ap, z = D^-1 r, fused (r,z), (r,r))", 8), ("blas1_direction", "ew_kernel<AypxDevOp> (p = z + beta p)", 3))
    out = []
    for k, (key, name, words) in enumerate(shapes):
        if key not in phase:
            out.append({"kernel": name, "fused_into": "spmv_pattern_fuse_kernel (x += alpha p_old and p = z + beta p_old ride on the SpMV's window fill: no launch of its own)",
                        "isolated_ms_per_launch": iso[k]["ms_per_launch"] if solver == "cg" and k < len(iso) else None})
            continue
        ms = phase[key]
        ach = words * 8 * n / (ms * 1e-3) / 1e9
        blk = {"kernel": name, "bound": "hbm", "bytes_per_launch": words * 8 * n, "ms_per_launch": ms, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
               "frac": ach / HBM_PEAK_GBS, "timed": "inside the solver's iterations (kryst_phase_timing, hipEvents between the phases of the phase run)"}
        if solver == "cg" and k < len(iso):
            blk["isolated_ms_per_launch"] = iso[k]["ms_per_launch"]
        out.append(blk)
    return out


class env_override:
    """Tuning knobs for a block of measurements.  The library reads them per entry point (per solve / session step inside a solver), so the
    block must not overlap with another setting: with several rank threads in one process the leader sets the variables between two
    barriers (nobody is inside the library then -- setenv must not race with a getenv)."""
    def __init__(self, group=None, **kv): self.g, self.kv, self.old = group, kv, {}
    def _sync(self):
        if self.g is not None and isinstance(self.g, ThreadedGroup): self.g.barrier()
    def __enter__(self):
        self._sync()
        if self.g is None or self.g.env_leader:
            for k, v in self.kv.items():
                self.old[k] = os.environ.get(k)
                os.environ[k] = v
        self._sync()
    def __exit__(self, *a):
        self._sync()
        if self.g is None or self.g.env_leader:
            for k, v in self.old.items():
                if v is None: os.environ.pop(k, None)
                else: os.environ[k] = v
        self._sync()


def stepped(K, ctx, group, method, a, pc, b, warmup, steps, batches=3):
    """W warm-up iterations, then `batches` batches of exactly K iterations of one stepping session (tol = 0), each bracketed by device
    synchronize + barrier on both sides, the maximum over ranks per batch; -> (headline batch seconds, stats, all batch seconds).  The headline
    is the MEAN of the batches when they agree within 10 % -- the steady-state rate: with x updated in batches of 8 iterations a window of K = 20
    iterations holds two or three batch passes, and the median would always pick a window with three -- and the MEDIAN when they do not: one
    disturbed batch must not become the headline."""
    def barrier():
        ctx.synchronize()
        group.barrier()
    x = ctx.vec(a.nrows())
    with K.Session(method, a, pc, b, x, tol=0.0, max_iters=warmup + batches * steps) as sess:
        sess.step(warmup)
        dts = []
        for _ in range(batches):
            barrier()
            t0 = time.perf_counter()
            sess.step(steps)
            barrier()
            dts.append(group.allreduce_max(time.perf_counter() - t0))
        stats = sess.end()
    assert stats.iterations == warmup + batches * steps, stats
    return batch_headline(dts), stats, dts


def batch_headline(dts):
    """Mean of the timed batches when the slowest is within 10 % of the fastest, else their median."""
    return sum(dts) / len(dts) if max(dts) <= 1.10 * min(dts) else sorted(dts)[len(dts) // 2]


GMRES_RESTART = 30


def timed_gmres(K, ctx, group, a, pc, b, warmup, steps, batches=3):
    """GmresSolver (gmres.rs:216-402; restart 30, the reference's default Left form with `pc`) has no stepping session -- its unit of
    work is the restart cycle -- so a batch is one device-resident SOLVE of exactly K iterations from x0 = 0 with tol = 0 (the iteration cap
    ends it: convergence.rs:25), after one warm-up solve of W iterations; every batch is bracketed by device synchronize + barrier on
    both sides, the maximum over ranks per batch; -> (median batch seconds, stats, all batch seconds).  The solve's own set-up (the
    initial residual, one small allocation) is inside the timed region, as SURVEY 8(d) words it: iterations / wall time of `solve`."""
    def barrier():
        ctx.synchronize()
        group.barrier()
    x = ctx.vec(a.nrows())
    if warmup > 0:
        K.GmresSolver(GMRES_RESTART, 0.0, warmup).solve(a, pc, b, x.fill(0.0))
    dts, st = [], None
    for _ in range(batches):
        x.fill(0.0)
        s = K.GmresSolver(GMRES_RESTART, 0.0, steps)
        barrier()
        t0 = time.perf_counter()
        st = s.solve(a, pc, b, x)
        barrier()
        dts.append(group.allreduce_max(time.perf_counter() - t0))
    assert st.iterations == steps, st
    return sorted(dts)[len(dts) // 2], st, dts


def measure(K, ctx, group, grid, solver, warmup, steps, phase_iters, batches=3, gmres_steps=0, provisional=None):
    """Everything measured on one grid size -> dict (the operator is built once; the plain-CSR figures re-run the same
    iterations with KRYST_SPMV_COMPRESS=0, which the library reads per session step)."""
    def barrier():
        ctx.synchronize()
        group.barrier()

    import contextlib
    settings = contextlib.ExitStack()                        # knobs the headline's winning form keeps for the rest of the measurements
    world, rank = group.world, group.rank
    stage(rank, f"create operator {grid}^3")
    a = K.CsrMatrix.stencil7(grid, "poisson", ctx=ctx)
    nloc = a.nrows()
    b = a.spmv(ctx.vec(nloc).fill(1.0))                      # b = A*1 (tests/preconditioner_integration.rs:25-31 convention)
    pc = K.Jacobi().setup(a) if solver in ("pcg", "gmres") else None

    def timed_iterations():
        if solver == "gmres":
            return timed_gmres(K, ctx, group, a, pc, b, warmup, steps, batches)
        return stepped(K, ctx, group, solver, a, pc, b, warmup, steps, batches)

    # N > 1: what the library chose by itself for this context and this operator (mailboxes / peer stores when their set-up and their checked
    # test reduction / test exchange succeed on every rank), then the conservative pair FIRST -- RCCL all-gather + grouped ncclSend / ncclRecv --
    # so that a measured line exists (`provisional`) before the forms that have never crossed xGMI are timed
    defaults = None
    if world > 1:
        defaults = {"scalar_reduce": ctx.scalar_reduce("query"), "halo": a.halo_mode("query") if hasattr(a, "halo_mode") else "rccl"}
        ctx.scalar_reduce("rccl")
        a.halo_mode("rccl")
    stage(rank, "warm-up + timed iterations" + (" (RCCL all-gather + RCCL halo)" if world > 1 else ""))
    dt, stats, dts = timed_iterations()
    if provisional is not None:
        provisional(dt, stats, a.encoding(), "RCCL all-gather + RCCL halo exchange only" if world > 1 else "headline iterations only")
    # N > 1: the same K iterations with the inner products crossing the ranks through the hipIpc mailboxes instead of two RCCL
    # all-gathers per iteration (kryst_ctx_scalar_reduce).  The faster path that reproduces the other's residual BIT FOR BIT is the
    # headline; both figures are reported.  (Every rank takes the same decision: times are max-reduced, scalars are identical.)
    reduce_info = None
    if world > 1:
        stage(rank, "mailbox setup + timed iterations (scalar all-reduce through hipIpc mailboxes)")
        reduce_info = {"path": "rccl", "value_rccl": steps / dt, "value_ipc": None, "ipc": "unavailable on this node (hipIpc export / mapping / test reduction failed)",
                       "library_defaults": defaults}
        if ctx.scalar_reduce("ipc") == "ipc":
            try:
                dt_ipc, stats_ipc, dts_ipc = timed_iterations()
                failed = 0.0
            except Exception as e:                  # e.g. a peer's stamp never arrived (KRYST_ERR_RCCL after the poll budget)
                dt_ipc, stats_ipc, dts_ipc, failed = float("inf"), None, None, 1.0
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
                    dt, stats, dts = dt_ipc, stats_ipc, dts_ipc
                    reduce_info["path"] = "ipc"
                else:
                    ctx.scalar_reduce("rccl")
    # N > 1: the forms of the halo exchange (kryst_ctx_halo_mode and KRYST_HALO_EARLY): the same bits; the fastest one is the headline, every
    # figure is reported
    if world > 1:
        stage(rank, "timed iterations (other halo forms)")
        forms = {"early": steps / dt}
        best = "early"
        with env_override(group, KRYST_HALO_EARLY="0"):
            dt_late, stats_late, dts_late = timed_iterations()
        same = group.allreduce_max(0.0 if stats_late.final_residual == stats.final_residual else 1.0) == 0.0
        forms["at_spmv"] = steps / dt_late
        late_wins = same and dt_late < dt
        if late_wins:
            dt, stats, dts, best = dt_late, stats_late, dts_late, "at_spmv"
        peer_note = None
        if hasattr(a, "halo_mode"):
            if a.halo_mode("peer") == "peer":
                try:
                    dt_p, stats_p, dts_p = timed_iterations()
                    failed = 0.0
                except Exception as e:
                    dt_p, stats_p, dts_p, failed = float("inf"), None, None, 1.0
                    peer_note = f"failed: {e}"
                if group.allreduce_max(failed) > 0.0:
                    peer_note = peer_note or "failed on another rank"
                    a.halo_mode("rccl")
                else:
                    same_p = group.allreduce_max(0.0 if stats_p.final_residual == stats.final_residual else 1.0) == 0.0
                    forms["peer_stores"] = steps / dt_p
                    peer_note = "bit-identical residual" if same_p else "DIFFERENT residual: not used"
                    if same_p and dt_p < dt:
                        dt, stats, dts, best = dt_p, stats_p, dts_p, "peer_stores"
                    else:
                        a.halo_mode("rccl")
            else:
                peer_note = "unavailable on this node (hipIpc export / mapping / test exchange of the halo buffers failed)"
        reduce_info.update(value_halo_early=forms["early"], value_halo_at_spmv=forms["at_spmv"], value_halo_peer_stores=forms.get("peer_stores"),
                           halo={"early": "early", "at_spmv": "at the SpMV", "peer_stores": "peer stores"}[best], peer_stores=peer_note)
        if best == "at_spmv":                                  # (the plain-CSR and phase runs below use the headline's setting)
            settings.enter_context(env_override(group, KRYST_HALO_EARLY="0"))
    enc = a.encoding()
    stage(rank, "timed iterations (plain CSR)")
    with env_override(group, KRYST_SPMV_COMPRESS="0"):
        dt_plain, stats_plain, dts_plain = timed_iterations()
    # the dominant kernel, timed live with HIP events on the compute stream (kryst_bench_spmv), in both storage forms
    stage(rank, "kernel timings")
    y = ctx.vec(nloc)
    # (three averages of 20 / 10 back-to-back launches each, the median of the three: one disturbed batch -- seen once, a 7x outlier --
    # must not become the roofline figure)
    spmv_ms = sorted(a.bench_spmv(b, y, fused_dots=1, reps=20) for _ in range(3))[1]
    with env_override(group, KRYST_SPMV_COMPRESS="0"):
        plain_ms = sorted(a.bench_spmv(b, y, fused_dots=1, reps=10) for _ in range(3))[1]
    # the plain kernel's traffic without its arithmetic on the same arrays, in the same process: what this mix of streams can reach here
    skeleton_ms = None
    if world == 1:
        try:
            skeleton_ms = sorted(a.bench_csr_skeleton(b, y, reps=10) for _ in range(3))[1]
        except Exception as e:
            sys.stderr.write(f"bench.py: stream skeleton skipped: {e}\n")
    # context: the device-copy rate at this footprint (hipMemcpy D2D of one vector, read + write)
    y.copy_from(b); ctx.synchronize()
    ctx.timer_start()
    for _ in range(10):
        y.copy_from(b)
    copy_gbs = 10 * 16.0 * nloc / (ctx.timer_stop() * 1e-3) / 1e9
    # where an iteration's device time goes (a separate short run with event marks between the phases)
    phases = None
    if phase_iters > 0:
        stage(rank, "phase timing")
        x = ctx.vec(nloc)
        if solver == "gmres":
            barrier()
            ctx.phase_timing_begin()
            K.GmresSolver(GMRES_RESTART, 0.0, phase_iters).solve(a, pc, b, x)
            ph = ctx.phase_timing_end()
        else:
            with K.Session(solver, a, pc, b, x, tol=0.0, max_iters=phase_iters + 2) as sess:
                sess.step(2)
                barrier()
                ctx.phase_timing_begin()
                sess.step(phase_iters)
                ph = ctx.phase_timing_end()
                sess.end()
        mine = {k: v / phase_iters for k, v in ph.items() if v > 0.0}
        mine["total"] = sum(mine.values())
        # "blas1" = every vector update of the iteration (its residual and direction passes are listed on their own as well)
        mine["blas1"] = mine.get("blas1", 0.0) + mine.get("blas1_residual", 0.0) + mine.get("blas1_direction", 0.0) + mine.get("blas1_xbatch", 0.0)
        phases = group.gather(mine)
    # N > 1: what one scalar collective costs end to end (local value -> RCCL all-gather -> rank-ordered fold -> host), so that the
    # first real multi-GPU run says how much of an iteration the two inner-product exchanges can be at most
    collective_us = None
    if world > 1:
        stage(rank, "scalar all-reduce round trips")
        for _ in range(10):
            ctx.all_reduce(1.0)
        barrier()
        t0 = time.perf_counter()
        for _ in range(100):
            ctx.all_reduce(1.0)
        collective_us = group.allreduce_max((time.perf_counter() - t0) / 100 * 1e6)
    # north_star: "CG/GMRES iterations/sec ... at 1, 2, 4 and 8 GPUs" -- GMRES(30) + Jacobi on the same system, same partition, with the
    # headline's scalar-reduce path and halo form (whole restart cycles: an iteration's cost grows with its position in the cycle)
    gm = None
    if gmres_steps > 0 and solver != "gmres":
        stage(rank, "GMRES(30) + Jacobi timed solves")
        try:
            pcj = pc if solver == "pcg" else K.Jacobi().setup(a)
            dt_g, st_g, dts_g = timed_gmres(K, ctx, group, a, pcj, b, min(warmup, GMRES_RESTART), gmres_steps, 2)
            gm = {"workload": f"gmres30_left_jacobi_poisson7_{grid}^3", "value": gmres_steps / dt_g, "unit": "iterations/s", "steps": gmres_steps,
                  "restart": GMRES_RESTART, "ms_per_step": dt_g / gmres_steps * 1e3, "batch_ms": [d * 1e3 for d in dts_g],
                  "final_residual": st_g.final_residual, "preconditioning": "Left + Jacobi, the reference's default form (gmres.rs:279-307)",
                  "rule": "one warm-up solve, then 2 solves of exactly `steps` iterations from x0 = 0 (tol 0), each bracketed by device synchronize + "
                          "barrier, max over ranks; value = steps / the later-sorted batch",
                  "algorithmic_bytes_per_iteration": spmv_bytes(nloc, a.nnz) + 1288 * nloc}
            del pcj
        except Exception as e:                              # the headline must survive a failing side measurement -- on every rank alike
            gm = {"error": f"{type(e).__name__}: {e}"}
        if world > 1 and group.allreduce_max(1.0 if "error" in gm else 0.0) > 0.0 and "error" not in gm:
            gm = {"error": "failed on another rank"}
    nnz_loc = a.nnz
    roof_csr = roofline_csr_of(grid, nloc, nnz_loc, plain_ms, world)
    fused_block = None
    mine_ph = phases[rank] if phases else None
    if solver in ("cg", "pcg") and mine_ph and "blas1_direction" not in mine_ph and "spmv" in mine_ph and enc[0] == "csr-p16":
        # the timed loop's SpMV launch IS the fused kernel: z, p_old [, x] read; p_new [, x], y written; one 16-bit pattern id per row.  With x updated
        # in batches (phase "blas1_xbatch" present) x is not touched here: XBatchOp applies x += alpha_i p_i for m iterations in one pass.
        xbatched = "blas1_xbatch" in mine_ph
        moved = (34 if xbatched else 50) * nloc
        ms_f = mine_ph["spmv"]
        ach = moved / (ms_f * 1e-3) / 1e9
        fused_block = {"bound": "hbm", "kernel": "spmv_pattern_fuse_kernel<1> (CSR-P16 staged window; the window fill forms p = z + beta p_old and stores p for its own rows"
                                                 + ("" if xbatched else ", carries the deferred x += alpha p_old") + ": direction pass + SpMV + (p,Ap) partials in one launch)",
                       "bytes_per_launch": moved, "bytes_model": "per row: z, p_old read (16 B), p_new, y written (16 B), pattern id (2 B)" if xbatched else
                                                                 "per row: z, p_old, x read (24 B), p_new, x, y written (24 B), pattern id (2 B)",
                       "ms_per_launch": ms_f, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                       "timed": "inside the solver's iterations (kryst_phase_timing: the `spmv` phase of the phase run)"}
        if xbatched:
            m = int(os.environ.get("KRYST_CG_X_BATCH", "8" if solver == "cg" else "7"))
            launches = max(1, len([i for i in range(3, phase_iters + 3) if i % m == 0]))      # the phase run covers iterations 3 .. phase_iters + 2
            ms_x = mine_ph["blas1_xbatch"] * phase_iters / launches
            fused_block["x_batch"] = {"kernel": f"ew_kernel<XBatchOp> (x += alpha_i p_i for {m} iterations in one pass: x read and written once, {m} direction vectors read)",
                                      "iterations_per_launch": m, "bytes_per_launch": (m + 2) * 8 * nloc, "ms_per_launch": ms_x,
                                      "achieved": (m + 2) * 8 * nloc / (ms_x * 1e-3) / 1e9, "unit": "GB/s", "frac": (m + 2) * 8 * nloc / (ms_x * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                      "ms_per_iteration": ms_x / m}
    if skeleton_ms:
        alg = spmv_bytes(nloc, nnz_loc)
        roof_csr["stream_skeleton"] = {
            "kernel": "csr_skeleton_kernel (the CSR arrays streamed, x read once, y written once: SURVEY 8(d)'s bytes, no gathers / products / row sums / fold; "
                      "same arrays, same process)", "ms_per_launch": skeleton_ms, "achieved": alg / (skeleton_ms * 1e-3) / 1e9, "unit": "GB/s",
            "frac": alg / (skeleton_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "kernel_over_skeleton": plain_ms / skeleton_ms}
    try:
        roof_csr["placement"] = dict(a.placement_info(), note="homes tried for (row_ptr, col, val) at creation (KRYST_CSR_PLACEMENT_TRIES; default 3 beyond 4 GB of CSR "
                                     "arrays), the traffic skeleton's ms per launch on each, the one kept")
    except Exception:
        pass
    settings.close()
    return {"dt": dt, "dts": dts, "dt_plain": dt_plain, "dts_plain": dts_plain, "stats": stats, "enc": enc, "nloc": nloc, "nnz_loc": nnz_loc,
            "collective_us": collective_us, "reduce_info": reduce_info, "gmres": gm,
            "roofline": dict(roofline_of(enc, grid, nloc, nnz_loc, spmv_ms, world, staged=a.pattern_info()["staged"]), fused_direction_spmv=fused_block),
            "roofline_csr": roof_csr,
            "blas1": blas1_in_solver(K, ctx, nloc, solver, phases[rank] if phases else None), "copy_gbs": copy_gbs, "phases": phases,
            "final_residual_plain": stats_plain.final_residual}


def timed_solve(K, ctx, make_solver, a, pc, b, repeat=2, want_x=False):
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
            best = (dt, st, x)
    return best if want_x else best[:2]


def true_relative_residual(K, ctx, a, b, x):
    """||b - A x|| / ||b|| with the library's own SpMV / norm (what a solver's own stopping quantity may or may not be)."""
    r = a.spmv(x)
    K.sub(b, r, r)
    return K.norm(r) / K.norm(b)


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


ILU_SETUP_NOTE = ("ilu_setup_first_ms = the first set-up of this operator in the process (code objects of the set-up kernels loaded, device memory for the "
                  "factors taken from the driver for the first time); ilu_setup_refactor_ms = set-ups of the SAME operator after the previous preconditioner "
                  "was destroyed -- what a caller who re-factors pays (Ilup::setup is called per matrix, ilup.rs:77-134): the destroyed preconditioner's "
                  "device blocks come back from the context's size-keyed pool (KRYST_DEV_POOL_MB, kryst_ctx_trim) instead of the driver, so the figure is "
                  "the factorisation and the blocked layout, not allocation")


def ilu_setup_times(K, ctx, a, refactors=2):
    """(first-call seconds, [re-factor seconds ...], preconditioner) of the true ILU(0) set-up: destroy, set up again, `refactors` times."""
    ctx.synchronize(); t0 = time.perf_counter()
    pc = K.TrueIlu0().setup(a)
    ctx.synchronize(); t_first = time.perf_counter() - t0
    again = []
    for _ in range(refactors):
        del pc
        ctx.synchronize(); t0 = time.perf_counter()
        pc = K.TrueIlu0().setup(a)
        ctx.synchronize(); again.append(time.perf_counter() - t0)
    return t_first, again, pc


def other_configs(K, ctx, steps, warmup):
    """BASELINE configs 3, 5 (256^3, solved to their tolerance), config 4's workload on one GPU (512^3 Jacobi-PCG, stepped) and the
    variable-coefficient operator at 256^3 / 512^3.  N = 1 only."""
    out = {}
    # ---- config 3: GMRES(30) Left + Jacobi on 256^3 convection-diffusion, tol 1e-8, max 600 (gmres.rs:216-402)
    a = K.CsrMatrix.stencil7(256, "convdiff", ctx=ctx); n = a.nrows()
    b = a.spmv(ctx.vec(n).fill(1.0))
    pc = K.Jacobi().setup(a)
    # the config as written (the reference's default Left arm) and, beside it, the reference's Right arm and the labelled textbook-Left
    # extension (precond_side 3): iterations, `converged` and the TRUE relative residual of each.  (Jacobi of this operator is a multiple of
    # the identity -- the diagonal is constant -- so a correct left or right preconditioning can only reproduce unpreconditioned GMRES(30).)
    forms = {}
    for label, side, rep in (("left_reference", K.Preconditioning.Left, 2), ("right_reference", K.Preconditioning.Right, 1),
                             ("left_textbook_extension", K.Preconditioning.LeftTextbook, 1)):
        dt_, st_, x_ = timed_solve(K, ctx, lambda: K.GmresSolver(30, 1e-8, 600).with_preconditioning(side), a, pc, b, repeat=rep, want_x=True)
        forms[label] = {"iterations": st_.iterations, "converged": bool(st_.converged), "final_residual": st_.final_residual,
                        "true_relative_residual": true_relative_residual(K, ctx, a, b, x_), "solve_seconds": dt_, "iterations_per_s": st_.iterations / dt_}
        if label == "left_reference":
            dt, st = dt_, st_
        del x_
    reached = [k for k, v in forms.items() if v["true_relative_residual"] <= 1e-8]
    out["config3_gmres30_jacobi_256"] = {
        "workload": "gmres30_left_jacobi_convdiff7_256^3", "value": st.iterations / dt, "unit": "iterations/s", "iterations": st.iterations,
        "converged": bool(st.converged), "final_residual": st.final_residual, "solve_seconds": dt, "spmv_encoding": a.encoding()[0],
        "forms": forms,
        "forms_note": ("left_reference = gmres.rs:240-247,279-307 as written (orthogonalises against an un-normalised Z[0]: stagnates; `converged` true "
                       "only means the iteration cap was hit, convergence.rs:25); right_reference = gmres.rs:248-260,308-342; left_textbook_extension = "
                       "precond_side 3, not in the reference.  "
                       + (f"Reached 1e-8 within 600 iterations: {', '.join(reached)}." if reached else
                          "NONE of the three reaches a true relative residual of 1e-8 within 600 iterations at restart 30 on this 256^3 operator: GMRES(30) itself "
                          "stagnates here whatever the (scalar) preconditioner; profiles/r05/config3_restart_sweep.jsonl holds the restart / iteration count that does")),
        "algorithmic_bytes_per_iteration": spmv_bytes(n, a.nnz) + 1288 * n,
        "roofline": stream_block(ctx, n, 0, "ew_kernel<MgsLinkOp> (Gram-Schmidt link z -= h v_i fused with the next link's dot: 3 reads + 1 write; "
                                 "31 links per iteration on average)", 4)}
    del pc, a, b
    # ---- config 5: right-preconditioned BiCGStab + true ILU(0) on 256^3 anisotropic Poisson, absolute tol 1e-8 ||b|| (bicgstab.rs:69-293)
    a = K.CsrMatrix.stencil7(256, "aniso", ctx=ctx); n = a.nrows()
    b = a.spmv(ctx.vec(n).fill(1.0)); bn = K.norm(b)
    t_first, t_again, pc = ilu_setup_times(K, ctx, a)
    dt, st = timed_solve(K, ctx, lambda: K.BiCgStabRightPcSolver(1e-8 * bn, 3000), a, pc, b)
    z = ctx.vec(n)
    out["config5_bicgstab_ilu0_256"] = {
        "workload": "bicgstab_right_true_ilu0_aniso7_256^3", "value": st.iterations / dt, "unit": "iterations/s", "iterations": st.iterations,
        "converged": bool(st.converged), "final_residual": st.final_residual, "solve_seconds": dt, "ilu_setup_first_ms": t_first * 1e3, "ilu_setup_refactor_ms": min(t_again) * 1e3, "ilu_setup_refactor_all_ms": [t * 1e3 for t in t_again],
        "ilu_setup_note": ILU_SETUP_NOTE,
        "spmv_encoding": a.encoding()[0], "note": "the reference's BiCGStab ignores pc (bicgstab.rs:70); the preconditioned form is a labelled extension",
        "roofline": tri_roofline(pc, b, z, n, a.nnz)}
    del pc, a, b, z
    # ---- config 4's workload on ONE GPU: Jacobi-PCG on 512^3 Poisson (pcg.rs:114-222), fixed stepped iterations
    k4 = min(steps, 100)
    a = K.CsrMatrix.stencil7(512, "poisson", ctx=ctx); n = a.nrows()
    b = a.spmv(ctx.vec(n).fill(1.0))
    pc = K.Jacobi().setup(a)
    dt, st, dts4 = stepped(K, ctx, SoloGroup(), "pcg", a, pc, b, warmup, k4)
    out["config4_jacobi_pcg_512"] = {
        "workload": "jacobi_pcg_poisson7_512^3 on one GPU (the 8-way partition is the --gpus 8 run)", "value": k4 / dt, "unit": "iterations/s", "steps": k4,
        "ms_per_step": dt / k4 * 1e3, "batch_ms": [d * 1e3 for d in dts4], "final_residual": st.final_residual, "spmv_encoding": a.encoding()[0],
        "algorithmic_bytes_per_iteration": spmv_bytes(n, a.nnz) + 136 * n}
    del pc, a, b
    # ---- variable coefficients: no row patterns, no value dictionary, no repeating coefficient chunks
    for grid in (256, 512):
        kv = min(steps, 100)
        a = K.CsrMatrix.stencil7(grid, "varcoef", ctx=ctx); n = a.nrows(); nnz = a.nnz
        b = a.spmv(ctx.vec(n).fill(1.0)); y = ctx.vec(n)
        enc = a.encoding()
        blk = {"workload": f"variable-coefficient 7-point diffusion operator, {grid}^3 (kind varcoef: per-edge weights from splitmix64)", "spmv_encoding": enc[0]}
        for form, env in (("default", {}), ("plain_csr", {"KRYST_SPMV_COMPRESS": "0"})):
            with env_override(None, **env):
                dt, _, _ = stepped(K, ctx, SoloGroup(), "cg", a, None, b, warmup, kv)
                ms = sorted(a.bench_spmv(b, y, fused_dots=1, reps=10) for _ in range(3))[1]
                e = a.encoding()
            if form == "default":
                blk.update(value=kv / dt, unit="iterations/s", steps=kv, ms_per_step=dt / kv * 1e3, roofline=roofline_of(e, grid, n, nnz, ms, 1, "varcoef"))
            else:
                blk.update(value_plain_csr=kv / dt, ms_per_step_plain_csr=dt / kv * 1e3, roofline_csr=roofline_csr_of(grid, n, nnz, ms, 2))
        t_first, t_again, pc = ilu_setup_times(K, ctx, a)
        blk.update(ilu_setup_first_ms=t_first * 1e3, ilu_setup_refactor_ms=min(t_again) * 1e3, ilu_setup_refactor_all_ms=[t * 1e3 for t in t_again])
        blk["ilu_apply"] = tri_roofline(pc, b, y, n, nnz, reps=10)
        out[f"variable_coefficient_{grid}"] = blk
        del pc, a, b, y
    try:
        out["general_ilu"] = general_ilu(K, ctx)
    except Exception as e:                                  # (scipy missing, out of memory ...: the headline does not depend on it)
        out["general_ilu"] = {"skipped": repr(e)}
    return out


def general_ilu(K, ctx):
    """SURVEY 8 row f-2, operators that are NOT 7-point boxes: true ILU(0) of a 27-point stencil on 96^3 (box-stencil wavefront solve,
    tri_box.h) and Ilup(1) of the 7-point Poisson operator on 128^3 (row-pipelined host elimination; its 13-entry factors take the same
    kernels), and true ILU(0) of a random band matrix of 2 M rows (deep, narrow dependency levels).  Setup = the second setup of the same operator in
    this process; apply = HIP events around 10 (5) applies."""
    import numpy as np
    import scipy.sparse as sp
    out = {}
    N = 96
    one = sp.diags([np.ones(N - 1), np.ones(N), np.ones(N - 1)], [-1, 0, 1])
    m = (sp.identity(N ** 3) * 28.0 - sp.kron(one, sp.kron(one, one))).tocsr()
    m.sort_indices()
    n = m.shape[0]
    a = K.CsrMatrix.from_csr(n, n, m.indptr, m.indices, m.data, ctx=ctx)
    del m
    for name, a_, mk in (("true_ilu0_27pt_96", a, lambda: K.TrueIlu0()), ("ilup1_poisson7_128", None, lambda: K.Ilup(1))):
        if a_ is None:
            a_ = K.CsrMatrix.stencil7(128, "poisson", ctx=ctx)
        n = a_.nrows()
        mk().setup(a_)
        ctx.synchronize(); t0 = time.perf_counter()
        pc = mk().setup(a_)
        ctx.synchronize(); setup_ms = (time.perf_counter() - t0) * 1e3
        r = ctx.vec(n).fill_splitmix(3); z = ctx.vec(n)
        ms = min(pc.bench_apply(r, z, 10) for _ in range(2))
        info = pc.ilu_info()
        blk = {"rows": n, "nnz": int(a_.nnz), "setup_ms": setup_ms, "apply_ms": ms, "form": info["form"]}
        if info["form"].startswith("box"):
            ns = sum(info["streams"])                        # coefficient streams the two factors have (of 13 each): only those are streamed
            moved = (ns * 8 + 8 + 32) * n
            blk.update(bytes_moved=moved, streams=info["streams"], regular=info["regular"],
                       bytes_model=f"{ns} coefficient streams + divisor + 32 n (r, y read; y, z written)",
                       achieved=moved / (ms * 1e-3) / 1e9, unit="GB/s", peak=HBM_PEAK_GBS, frac=moved / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                       hyperplanes=info["levels"][0], us_per_hyperplane=ms * 1e3 / max(1, 2 * info["levels"][0]))
        if name == "true_ilu0_27pt_96":
            # the preconditioner inside a solve: right-preconditioned BiCGStab to 1e-10 ||b|| (bicgstab.rs:69-293 + the labelled right-pc extension)
            b = a_.spmv(ctx.vec(n).fill(1.0)); bn = K.norm(b)
            dt, st = timed_solve(K, ctx, lambda: K.BiCgStabRightPcSolver(1e-10 * bn, 500), a_, pc, b)
            blk["bicgstab_right_pc"] = {"iterations": st.iterations, "converged": bool(st.converged), "final_residual": st.final_residual, "solve_seconds": dt,
                                        "iterations_per_s": st.iterations / dt, "spmv_encoding": a_.encoding()[0]}
            del b
        out[name] = blk
        del pc, r, z, a_
    if os.environ.get("KRYST_BENCH_RANDOM_BAND", "1") != "0":
        # true ILU(0) of the random band matrix of tools/band_apply.py (2 M rows, 9 per row, |i - j| <= 2000: 10 716 dependency
        # levels of ~187 rows per factor) -- level-ordered factors, their narrow levels solved by the one-workgroup barrier-free kernel (tri_run_free_kernel)
        NR = 2000000
        rng = np.random.default_rng(1)
        rows = np.repeat(np.arange(NR), 9)
        cols = np.clip(rows + rng.integers(-2000, 2001, len(rows)), 0, NR - 1)
        m = sp.csr_matrix((rng.uniform(-1.0, 1.0, len(rows)), (rows, cols)), shape=(NR, NR))
        del rows, cols
        m.sum_duplicates()
        m = m - sp.diags(m.diagonal()) + sp.diags(np.asarray(abs(m).sum(axis=1)).ravel() + 1.0)
        m = m.tocsr(); m.sort_indices()
        a_ = K.CsrMatrix.from_csr(NR, NR, m.indptr, m.indices, m.data, ctx=ctx)
        nnz = int(m.nnz)
        del m
        K.TrueIlu0().setup(a_)
        ctx.synchronize(); t0 = time.perf_counter()
        pc = K.TrueIlu0().setup(a_)
        ctx.synchronize(); setup_ms = (time.perf_counter() - t0) * 1e3
        r = ctx.vec(NR).fill_splitmix(3); z = ctx.vec(NR)
        ms = min(pc.bench_apply(r, z, 5) for _ in range(2))
        info = pc.ilu_info()
        out["true_ilu0_random_band_2m"] = {"rows": NR, "nnz": nnz, "setup_ms": setup_ms, "apply_ms": ms, "form": info["form"], "levels_L_U": info["levels"],
                                           "us_per_level": ms * 1e3 / max(1, sum(info["levels"]))}
        del pc, r, z, a_
    return out



# ---------------------------------------------------------------------------------------------------------------- self-launch
def _free_port_range(width=10):
    """A local port with the next `width` ports free as well right now (the socket rendezvous listens on MASTER_PORT + 1 ...)."""
    import socket
    for _ in range(64):
        with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        if port + width >= 65000:
            continue
        ok = True
        for k in range(1, width):
            with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as t:
                try:
                    t.bind(("127.0.0.1", port + k))
                except OSError:
                    ok = False
                    break
        if ok:
            return port
    raise RuntimeError("bench.py: no free local port range for the rendezvous")


def self_launch(argv, procs, limit_s):
    """`python3 bench.py --gpus N` started PLAINLY (no WORLD_SIZE in the environment): this process becomes the launcher that
    `mpirun` is for the reference (src/parallel/mpi_comm.rs:49-55).  It starts `procs` FRESH child processes of this very file with
    RANK / WORLD_SIZE / LOCAL_RANK / MASTER_ADDR / MASTER_PORT set (one rank per GPU; the children rendezvous over
    kryst_amd/launch.py's sockets unless --launcher torch is given), relays rank 0's stdout -- the JSON line -- as its only stdout,
    lets every rank's stage markers through on stderr, ends the others when one child exits non-zero and returns that code.
    The launcher itself never touches the GPU: it imports neither kryst_amd nor torch and never execs (the children are started with
    subprocess.Popen)."""
    import signal
    import subprocess
    port = _free_port_range()
    here = os.path.abspath(__file__)
    kids = []
    err_fd = sys.stderr.fileno()
    for r in range(procs):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(procs), LOCAL_RANK=str(r), LOCAL_WORLD_SIZE=str(procs),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), KRYST_BENCH_SELF_LAUNCHED="1")
        env.setdefault("KRYST_LAUNCHER", "socket")            # no torch import in N processes unless asked for (--launcher torch)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        kids.append(subprocess.Popen([sys.executable, here] + list(argv), env=env, stdin=subprocess.DEVNULL,
                                     stdout=None if r == 0 else err_fd))     # rank 0's stdout IS this process's stdout
    sys.stderr.write(f"[bench launcher +{time.time() - _T0:7.2f}s] started {procs} rank process(es), MASTER_PORT {port}: pids "
                     + " ".join(str(k.pid) for k in kids) + "\n"); sys.stderr.flush()

    def end_all(sig):
        for k in kids:
            if k.poll() is None:
                try:
                    k.send_signal(sig)                        # (exactly the processes started above, by pid)
                except OSError:
                    pass

    rc, deadline = 0, time.time() + limit_s
    try:
        while True:
            codes = [k.poll() for k in kids]
            bad = [c for c in codes if c not in (None, 0)]
            if bad:
                rc = bad[0] if bad[0] > 0 else 128 - bad[0]
                sys.stderr.write(f"[bench launcher] rank process {codes.index(bad[0])} exited with {bad[0]}: ending the others\n"); sys.stderr.flush()
                break
            if all(c == 0 for c in codes):
                break
            if time.time() > deadline:
                sys.stderr.write(f"[bench launcher] {limit_s:.0f} s without every rank finishing: ending the ranks\n"); sys.stderr.flush()
                rc = 124
                break
            time.sleep(0.05)
    finally:
        end_all(signal.SIGTERM)
        t_end = time.time() + 5.0
        while any(k.poll() is None for k in kids) and time.time() < t_end:
            time.sleep(0.05)
        end_all(signal.SIGKILL)
        for k in kids:
            try:
                k.wait(timeout=5)
            except Exception:
                pass
    dump = os.environ.get("KRYST_BENCH_LAUNCHER_MAPS")        # tests: proof that the launcher mapped no HIP / kryst library
    if dump:
        with open(dump, "w") as f:
            f.write(open("/proc/self/maps").read())
    return rc


def rank_main(args, group, rank, world, dev, grid):
    """One rank's whole run (the process's only one, or one of its rank threads)."""
    import kryst_amd as K
    stage(rank, "comm init" if world > 1 else "context")
    if world > 1:
        uid = group.broadcast_bytes(K.Context.unique_id() if rank == 0 else None)
        ctx = K.Context(dev, rank, world, uid)
    else:
        ctx = K.Context(0)

    n = grid ** 3
    nnz = 7 * n - 6 * grid * grid
    wl = {"pcg": "jacobi_pcg", "gmres": "gmres30_left_jacobi"}.get(args.solver, "cg")

    def headline(dt, stats, enc):
        return {
            "metric": "gmres_iterations_per_sec" if args.solver == "gmres" else "cg_iterations_per_sec", "value": args.steps / dt, "unit": "iterations/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{wl}_poisson7_{grid}^3", "grid": grid, "rows": n, "nnz": nnz, "partition": f"{world} k-slab(s)",
                       "rhs": "A*1", "x0": "0", "final_residual": stats.final_residual, "spmv_encoding": enc[0],
                       "launcher": type(group).__name__ if not isinstance(group, ThreadedGroup) else f"{type(group.sh.inner).__name__} x {group.sh.per} rank threads per process"}}

    def provisional(dt, stats, enc, what):
        # every rank remembers that a measured line exists (the watchdog then ends the run with exit code 0); rank 0 holds the line itself
        line = headline(dt, stats, enc)
        line.update(roofline=None, provisional=what)
        _PROVISIONAL["line"] = line

    m = measure(K, ctx, group, grid, args.solver, args.warmup, args.steps, args.phase_iters, args.batches, args.gmres_steps, provisional)
    out = headline(m["dt"], m["stats"], m["enc"])
    roof = m["roofline"]
    rc = m["roofline_csr"]
    fb = roof.get("fused_direction_spmv")
    if fb:
        # The timed loop's dominant kernel is the FUSED kernel (direction pass + SpMV + (p,Ap) partials; the staged-window SpMV alone is not
        # launched by the loop at all): it is the top-level block, timed INSIDE the solver's iterations, with the HBM traffic of exactly that
        # kernel from two rocprofv3 --pmc passes over a CG session of this command (world == 1).  The SpMV-alone block moves to `spmv_alone`.
        alone = {k: v for k, v in roof.items() if k != "fused_direction_spmv"}
        top = dict(fb)
        if world == 1:
            tr = live_traffic(grid, "fused")
            if tr:
                top["traffic"] = tr
                top["traffic_source"] = "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over a CG session (tools/cg_only.py) run by this command, read side calibrated on DotOp"
                top["frac_traffic"] = tr / (top["ms_per_launch"] * 1e-3) / 1e9 / HBM_PEAK_GBS
        top.setdefault("traffic", None)
        top["share_of_iteration"] = top["ms_per_launch"] / (m["dt"] / args.steps * 1e3)
        top["spmv_alone"] = alone
        top["fused_direction_spmv"] = {k: v for k, v in fb.items()}          # (the same figures under the round's earlier key)
        roof = top
    # SURVEY 8(d)'s own pair INSIDE the block the driver keeps: the plain-CSR kernel (the only one that moves 8(d)'s 12 nnz + 4 (n + 1) + 16 n
    # bytes) priced at those bytes, and the iterations/s of the same K iterations with that kernel (`value_sec8d`)
    roof["sec8d"] = {"kernel": rc["kernel"], "frac": rc["frac"], "achieved": rc["achieved"], "unit": "GB/s", "peak": HBM_PEAK_GBS,
                     "bytes_per_launch": rc["bytes_per_launch"], "bytes_model": rc["bytes_model"], "ms_per_launch": rc["ms_per_launch"],
                     "traffic": rc.get("traffic"), "stream_skeleton": rc.get("stream_skeleton"), "placement": rc.get("placement"),
                     "value_sec8d": args.steps / m["dt_plain"], "value_unit": "iterations/s",
                     "note": "north_star's '% of HBM roofline on CSR SpMV' and the iterations/s that go with it; `roofline.frac` above prices the "
                             "default (lossless, more compact) storage form at ITS bytes"}
    out.update({
        "timing": {"batches": len(m["dts"]), "batch_ms": [d * 1e3 for d in m["dts"]], "batch_ms_plain_csr": [d * 1e3 for d in m["dts_plain"]],
                   "rule": "W warm-up iterations, then `batches` batches of exactly K iterations of ONE stepping session (GMRES: one solve of exactly K "
                           "iterations per batch), each bracketed by device synchronize + barrier on both sides, max over ranks per batch; value = K / the MEAN batch when "
                           "the batches agree within 10 % (x is updated in batches of 8 iterations: a window of K iterations holds a whole number of those "
                           "passes only on average), K / the MEDIAN batch when one of them is disturbed"},
        "value_sec8d": args.steps / m["dt_plain"],
        "value_plain_csr": args.steps / m["dt_plain"], "ms_per_step_plain_csr": m["dt_plain"] / args.steps * 1e3,
        "roofline": roof, "roofline_csr": m["roofline_csr"], "roofline_blas1": m["blas1"],
        "gmres30_jacobi": m["gmres"],    # north_star: CG / GMRES iterations/s at 1, 2, 4, 8 GPUs -- GMRES(30) + Jacobi on the same system and partition
        "measured_copy_GBs": m["copy_gbs"],
        "phase_ms": m["phases"],         # per rank: device ms per iteration by phase (spmv / halo_wait / spmv_boundary / reduce / blas1 = blas1_residual + blas1_direction + other)
        "scalar_all_reduce_us": m["collective_us"],   # N > 1: host round trip of one scalar all-reduce (all-gather + ordered fold + sync), max over ranks
        "scalar_reduce": m["reduce_info"],            # N > 1: which path carried the inner products in `value` (RCCL all-gather or hipIpc mailboxes) and both figures
    })
    _PROVISIONAL["line"] = dict(out, provisional="headline measurements complete; the side measurements (256^3, other configs, CPU baseline) were cut short")
    base256 = None
    if world == 1 and grid != 256 and not args.no_256:
        m2 = measure(K, ctx, group, 256, args.solver, args.warmup, args.steps, args.phase_iters, args.batches, args.gmres_steps)
        out["config1_256"] = {"workload": f"{wl}_poisson7_256^3", "value": args.steps / m2["dt"], "unit": "iterations/s",
                              "ms_per_step": m2["dt"] / args.steps * 1e3, "batch_ms": [d * 1e3 for d in m2["dts"]], "value_plain_csr": args.steps / m2["dt_plain"],
                              "value_sec8d": args.steps / m2["dt_plain"],
                              "roofline": m2["roofline"], "roofline_csr": m2["roofline_csr"], "roofline_blas1": m2["blas1"],
                              "gmres30_jacobi": m2["gmres"], "phase_ms": m2["phases"]}
        if not args.no_cpu_baseline:
            stage(rank, "cpu baseline + full-size parity 256^3")
            try:
                base256 = cpu_cg(256, 8.0, ctx)
                out["config1_256"]["cpu_baseline"] = base256
            except Exception as e:
                out["config1_256"]["cpu_baseline"] = {"value": None, "unit": "cg_iterations/s", "cores": 0, "kind": "port", "sample": f"failed: {e}"}
    if world == 1 and not args.no_configs:
        stage(rank, "other BASELINE configs")
        try:
            out.update(other_configs(K, ctx, args.steps, args.warmup))
        except Exception as e:                              # the headline line must survive a failing side measurement
            out["other_configs_error"] = f"{type(e).__name__}: {e}"
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        stage(rank, f"cpu baseline + full-size parity {grid}^3")
        out["cpu_baseline"] = cpu_baseline(grid, base256, ctx)
        # the oracle iterations of the baseline double as the full-size parity check (north_star: residuals matching the CPU reference)
        par = [b["parity_at_size"] for b in (out["cpu_baseline"], base256 or {}) if isinstance(b, dict) and "parity_at_size" in b]
        if par:
            out["parity_at_size"] = par
    stage(rank, "gather / print")
    _PROVISIONAL.clear()                       # (the real line follows: the watchdog must not print a second one)
    if rank == 0:
        print(json.dumps(out), flush=True)
    group.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--batches", type=int, default=3, help="timed batches of --steps iterations each; their mean is the headline (the median when one batch is disturbed)")
    ap.add_argument("--grid", type=int, default=0, help="grid edge (default 512 for every N)")
    ap.add_argument("--solver", default="cg", choices=["cg", "pcg", "gmres"],
                    help="headline iteration: cg (cg.rs), pcg = Jacobi-PCG (pcg.rs), gmres = GMRES(30) Left + Jacobi (gmres.rs; a batch is one solve of K iterations)")
    ap.add_argument("--gmres-steps", type=int, default=60,
                    help="iterations of the GMRES(30) + Jacobi block measured beside a cg / pcg headline (whole restart cycles; 0: skip)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-256", action="store_true", help="skip the config1_256 block (N = 1 measures BASELINE configs[1]'s 256^3 grid too)")
    ap.add_argument("--no-configs", action="store_true", help="skip the other BASELINE configs and the variable-coefficient blocks (N = 1)")
    ap.add_argument("--phase-iters", type=int, default=56, help="iterations of the per-phase timing run (0: skip)")
    ap.add_argument("--launcher", default="auto", choices=["auto", "torch", "socket"],
                    help="N > 1 plumbing for the RCCL id / barrier: torch.distributed gloo (default) or kryst_amd/launch.py (no torch)")
    ap.add_argument("--ranks-per-process", type=int, default=1,
                    help="REHEARSAL ONLY (one-GPU box, KRYST_BENCH_DEVICE): this many ranks per launched process, a host thread each -- 8 ranks as "
                         "4 processes x 2, because a GPU box admits at most 6 processes on its card; --gpus then counts ranks")
    args = ap.parse_args()

    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: what RCCL needs between processes on this driver
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # started plainly (the way the driver starts --gpus 1): become the launcher -- before anything of this process touches the GPU
        per = max(1, args.ranks_per_process)
        if args.gpus % per:
            sys.exit(f"bench.py: --gpus {args.gpus} is not a multiple of --ranks-per-process {per}")
        limit = float(os.environ.get("KRYST_BENCH_WATCHDOG_S", "280")) + 15.0
        sys.exit(self_launch(sys.argv[1:], args.gpus // per, limit))
    procs = int(os.environ.get("WORLD_SIZE", "1"))
    prank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    per = max(1, args.ranks_per_process)
    world = procs * per

    # A hung collective must not hang the node, and the driver must learn WHERE it hung: the watchdog fires BELOW the driver's own
    # timeout (600 s in round 3) and names every local rank's last stage marker.
    limit = float(os.environ.get("KRYST_BENCH_WATCHDOG_S", "280" if world > 1 else "560"))
    def _watchdog():
        where = "; ".join(f"rank {r}: {nm}" for r, nm in sorted(_STAGE.items())) or "before the first stage"
        sys.stderr.write(f"bench.py: watchdog timeout ({limit:.0f} s) -- last stage markers: {where}\n"); sys.stderr.flush()
        line = _PROVISIONAL.get("line")
        if line is None:
            os._exit(124)
        # a complete measurement exists and only a LATER stage hung: the measured line is printed (marked) instead of being lost with the run.
        # Every rank reaches this branch alike (the provisional line is recorded right after a collective), so every rank exits with 0.
        if prank == 0:
            line = dict(line, truncated_by_watchdog=f"{limit:.0f} s; last stage markers: {where}")
            sys.stdout.write(json.dumps(line) + "\n"); sys.stdout.flush()
        os._exit(0)
    wd = threading.Timer(limit, _watchdog); wd.daemon = True; wd.start()

    if args.gpus != world:
        args.gpus = world
    grid = args.grid or 512          # ONE fixed problem for every N (strong scaling, north_star): 512^3

    stage(prank * per, f"rendezvous (process {prank} of {procs}, {per} rank(s) per process, LOCAL_RANK {local_rank})")
    inner = make_group(args.launcher, prank, procs)
    # KRYST_BENCH_DEVICE: rehearsal of the N > 1 flow on a one-GPU box (all ranks on one device, RCCL stand-in)
    dev = int(os.environ.get("KRYST_BENCH_DEVICE", local_rank))
    if world > 1:
        from kryst_amd._ffi import device_count
        ndev = device_count()
        if dev >= ndev:                  # one rank per GPU: a rank without a device of its own must fail loudly, not share one silently
            sys.stderr.write(f"bench.py: rank {prank} has LOCAL_RANK {local_rank} -> device {dev}, but this process sees {ndev} device(s) "
                             "(HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES?); one GPU per rank is required\n")
            sys.stderr.flush()
            os._exit(2)
    if per == 1:
        return rank_main(args, inner, prank, world, dev, grid)
    shared = ThreadedGroup.Shared(inner, per)
    failed = []

    def guarded(t):
        try:
            rank_main(args, ThreadedGroup(shared, t), prank * per + t, world, dev, grid)
        except BaseException:                       # noqa: BLE001  (a sibling thread would wait for this rank inside a collective for ever)
            import traceback
            traceback.print_exc()
            sys.stderr.flush()
            failed.append(t)
            os._exit(3)

    ts = [threading.Thread(target=guarded, args=(t,)) for t in range(per)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()


if __name__ == "__main__":
    main()
