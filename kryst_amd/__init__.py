"""kryst_amd -- host-side mirror of kryst's operator / preconditioner / solver interface for the MI355X path.

The classes keep the reference's names, constructor arguments, builder methods, public fields and error
behaviour (paths relative to the kryst crate):

    CsrMatrix.from_csr(nrows, ncols, row_ptr, col_idx, values)      src/matrix/sparse.rs:28-46
    CsrMatrix.spmv(x, y) / matvec(x, y)                             src/matrix/sparse.rs:56-67, core/traits.rs:4-7
    dot(x, y), norm(x)                                              src/core/wrappers.rs:90-127
    Jacobi / Ilu0 / Ilup / Chebyshev  .setup(a) .apply(r, z)        src/preconditioner/*.rs
    apply_chebyshev(a, r, z, alpha, beta, m)                        src/preconditioner/chebyshev.rs:83-140
    CgSolver / PcgSolver / GmresSolver / BiCgStabSolver .solve(a, pc, b, x) -> SolveStats   src/solver/*.rs
    Convergence, SolveStats, KError, CgNormType, Preconditioning    src/utils/convergence.rs, src/error.rs

Everything executes in libkryst_hip.so (hand-written HIP for gfx950) through the C ABI of include/kryst_hip.h.
There is no CPU fallback and no torch dependency.
"""
import ctypes as C
import enum
import numpy as np

from . import _ffi
from ._ffi import KError, lib, check

__all__ = ["Context", "DeviceVec", "CsrMatrix", "dot", "norm", "Jacobi", "Ilu0", "Ilup", "Ilut", "TrueIlu0", "Chebyshev",
           "ChebyshevPc", "IdentityPc", "ApproxInv", "apply_chebyshev", "Convergence", "SolveStats", "CgNormType",
           "Preconditioning", "CgSolver", "PcgSolver", "GmresSolver", "FgmresSolver", "Orthog", "CgsSolver", "TfqmrSolver", "BiCgStabSolver", "BiCgStabRightPcSolver", "Session", "KspContext", "SolverKind", "PC", "KError", "reduce_spec",
           "host_stencil7", "partition_rows", "halo_recv_plan", "read_matrix_market", "read_petsc_binary", "host_ilup", "host_ilut", "host_levels"]


def _dp(a):
    return a.ctypes.data_as(_ffi.c_dp)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def reduce_spec():
    """(T, V, F) of the library's fixed inner-product tree (kryst_reduce_spec)."""
    t, v, f = C.c_int32(), C.c_int32(), C.c_int32()
    lib().kryst_reduce_spec(C.byref(t), C.byref(v), C.byref(f))
    return t.value, v.value, f.value


class Context:
    """One GPU (one rank).  Replaces the Comm objects of src/parallel (RayonComm / MpiComm)."""

    _default = None

    def __init__(self, device=0, rank=0, nranks=1, unique_id=None):
        self.h = _ffi.Handle()
        if nranks == 1 and unique_id is None:
            check(lib().kryst_ctx_create(device, C.byref(self.h)))
        else:
            buf = C.create_string_buffer(bytes(unique_id), 128)
            check(lib().kryst_ctx_create_dist(device, rank, nranks, buf, C.byref(self.h)))
        self.rank, self.nranks, self.device = rank, nranks, device

    @staticmethod
    def default():
        if Context._default is None:
            Context._default = Context(0)
        return Context._default

    @staticmethod
    def unique_id():
        buf = C.create_string_buffer(128)
        check(lib().kryst_comm_unique_id(buf))
        return buf.raw

    def size(self):
        return self.nranks

    def barrier(self):
        check(lib().kryst_comm_barrier(self.h))

    def all_reduce(self, x):
        out = C.c_double()
        check(lib().kryst_comm_all_reduce(self.h, float(x), C.byref(out)))
        return out.value

    def scalar_reduce(self, mode):
        """'rccl' | 'ipc' | 'query': how the solvers' inner products cross the ranks (kryst_ctx_scalar_reduce; collective except 'query').
        Returns the mode in use afterwards ('ipc' falls back to 'rccl' on every rank when a mailbox cannot be mapped or the test reduction
        does not arrive intact).  A context of several ranks starts on 'ipc' when that works everywhere (KRYST_SCALAR_REDUCE=rccl: never)."""
        active = C.c_int32(0)
        rc = lib().kryst_ctx_scalar_reduce(self.h, {"rccl": 0, "ipc": 1, "query": -1}[mode], C.byref(active))
        if rc not in (0, 6):
            check(rc)
        return "ipc" if active.value else "rccl"

    def synchronize(self):
        check(lib().kryst_ctx_synchronize(self.h))

    def trim(self):
        """Give the device blocks kept for reuse (destroyed ILU preconditioners' storage, the solvers' work arena) back to the driver
        (kryst_ctx_trim) -> bytes released."""
        n = C.c_int64(0)
        check(lib().kryst_ctx_trim(self.h, C.byref(n)))
        return n.value

    def poison_lds(self):
        """Test hook: NaNs into every compute unit's LDS (a kernel must not depend on what LDS held before it started)."""
        check(lib().kryst_bench_poison_lds(self.h))

    def timer_start(self):
        check(lib().kryst_ctx_timer_start(self.h))

    def timer_stop(self):
        ms = C.c_double()
        check(lib().kryst_ctx_timer_stop(self.h, C.byref(ms)))
        return ms.value

    def phase_timing_begin(self):
        """Start charging device time to phases (spmv / halo_wait / spmv_boundary / reduce / blas1 / pc): measurement only."""
        check(lib().kryst_phase_timing_begin(self.h))

    def phase_timing_end(self):
        """-> {phase name: ms} of the work enqueued since phase_timing_begin (synchronises the context)."""
        n = lib().kryst_phase_count()
        ms = (C.c_double * n)()
        check(lib().kryst_phase_timing_end(self.h, ms, n))
        return {lib().kryst_phase_name(i).decode(): ms[i] for i in range(n)}

    def vec(self, n_or_array):
        return DeviceVec(self, n_or_array)

    def close(self):
        if self.h:
            lib().kryst_ctx_destroy(self.h)
            self.h = None


class DeviceVec:
    """A Vec<f64> resident in HBM."""

    def __init__(self, ctx, n_or_array):
        self.ctx = ctx
        self.h = _ffi.Handle()
        if np.isscalar(n_or_array):
            self.n = int(n_or_array)
            check(lib().kryst_vec_create(ctx.h, self.n, C.byref(self.h)))
        else:
            a = _f64(n_or_array)
            self.n = len(a)
            check(lib().kryst_vec_create(ctx.h, self.n, C.byref(self.h)))
            self.upload(a)

    def __len__(self):
        return self.n

    def upload(self, a):
        a = _f64(a)
        check(lib().kryst_vec_upload(self.h, _dp(a), len(a)))
        return self

    def to_host(self):
        out = np.empty(self.n)
        check(lib().kryst_vec_download(self.h, _dp(out), self.n))
        return out

    def fill(self, v):
        check(lib().kryst_vec_fill(self.h, float(v)))
        return self

    def fill_splitmix(self, seed, global_offset=0):
        check(lib().kryst_vec_fill_splitmix(self.h, seed, global_offset))
        return self

    def copy_from(self, other):
        check(lib().kryst_vec_copy(self.h, other.h))
        return self

    def __del__(self):
        try:
            if self.h and self.ctx.h:
                lib().kryst_vec_destroy(self.h)
        except Exception:
            pass


def dot(x, y):
    """InnerProduct::dot (wrappers.rs:90-108) on device vectors."""
    out = C.c_double()
    check(lib().kryst_dot(x.h, y.h, C.byref(out)))
    return out.value


def norm(x):
    """InnerProduct::norm (wrappers.rs:110-127)."""
    out = C.c_double()
    check(lib().kryst_norm(x.h, C.byref(out)))
    return out.value


def axpy(alpha, x, y):
    check(lib().kryst_axpy(float(alpha), x.h, y.h))


def aypx(beta, x, y):
    check(lib().kryst_aypx(float(beta), x.h, y.h))


def sub(a, b, out):
    """out[i] = a[i] - b[i] (cg.rs:123 `bi - ax`); `out` may be `a` or `b`."""
    check(lib().kryst_sub(a.h, b.h, out.h))
    return out


STENCIL_KINDS = {"poisson": 0, "aniso": 1, "convdiff": 2, "varcoef": 3}   # kryst_csr_create_stencil7 / kryst_host_stencil7


class CsrMatrix:
    """CsrMatrix<f64> (src/matrix/sparse.rs:22-46) living on the GPU; implements SparseMatrix::spmv and MatVec."""

    def __init__(self, ctx, handle):
        self.ctx, self.h = ctx, handle
        nr, nc, nz = C.c_int64(), C.c_int64(), C.c_int64()
        check(lib().kryst_csr_shape(self.h, C.byref(nr), C.byref(nc), C.byref(nz)))
        self._nrows, self._ncols, self.nnz = nr.value, nc.value, nz.value

    @staticmethod
    def from_csr(nrows, ncols, row_ptr, col_idx, values, ctx=None):
        """from_csr(nrows, ncols, row_ptr: Vec<usize>, col_idx: Vec<usize>, values)  sparse.rs:28-46.
        Violating new_checked's preconditions raises KError (the reference panics)."""
        ctx = ctx or Context.default()
        rp = np.ascontiguousarray(row_ptr, dtype=np.uint64)
        ci = np.ascontiguousarray(col_idx, dtype=np.uint64)
        va = _f64(values)
        if len(rp) != nrows + 1 or len(ci) != len(va) or (len(rp) and int(rp[-1]) != len(va)):
            raise KError(102, "from_csr: inconsistent array lengths")
        h = _ffi.Handle()
        check(lib().kryst_csr_create(ctx.h, nrows, ncols, rp.ctypes.data_as(_ffi.c_u64p), ci.ctypes.data_as(_ffi.c_u64p),
                                     _dp(va), C.byref(h)))
        return CsrMatrix(ctx, h)

    @staticmethod
    def from_csr_i32(nrows, ncols, row_ptr, col_idx, values, ctx=None):
        ctx = ctx or Context.default()
        rp = np.ascontiguousarray(row_ptr, dtype=np.int64)
        ci = np.ascontiguousarray(col_idx, dtype=np.int32)
        va = _f64(values)
        h = _ffi.Handle()
        check(lib().kryst_csr_create_i32(ctx.h, nrows, ncols, rp.ctypes.data_as(_ffi.c_i64p),
                                         ci.ctypes.data_as(_ffi.c_i32p), _dp(va), C.byref(h)))
        return CsrMatrix(ctx, h)

    @staticmethod
    def from_csr_dist(ctx, n_global, row_offsets, row_ptr, col_idx_global, values):
        """Row block [row_offsets[rank], row_offsets[rank+1]) of a row-partitioned operator (global columns)."""
        ro = np.ascontiguousarray(row_offsets, dtype=np.int64)
        rp = np.ascontiguousarray(row_ptr, dtype=np.int64)
        ci = np.ascontiguousarray(col_idx_global, dtype=np.int64)
        va = _f64(values)
        h = _ffi.Handle()
        check(lib().kryst_csr_create_dist(ctx.h, n_global, ro.ctypes.data_as(_ffi.c_i64p), rp.ctypes.data_as(_ffi.c_i64p),
                                          ci.ctypes.data_as(_ffi.c_i64p), _dp(va), C.byref(h)))
        return CsrMatrix(ctx, h)

    @staticmethod
    def from_matrix_market(path, ctx=None):
        nr, nc, rp, ci, va = read_matrix_market(path)
        return CsrMatrix.from_csr(nr, nc, rp, ci, va, ctx=ctx)

    @staticmethod
    def from_petsc_binary(path, ctx=None):
        nr, nc, rp, ci, va = read_petsc_binary(path)
        return CsrMatrix.from_csr(nr, nc, rp, ci, va, ctx=ctx)

    @staticmethod
    def stencil7(N, kind="poisson", ctx=None):
        """Synthetic 7-point operator on an N^3 grid; each rank of a distributed ctx gets its k-slab."""
        ctx = ctx or Context.default()
        h = _ffi.Handle()
        check(lib().kryst_csr_create_stencil7(ctx.h, N, STENCIL_KINDS[kind], C.byref(h)))
        return CsrMatrix(ctx, h)

    def nrows(self):
        return self._nrows

    def ncols(self):
        return self._ncols

    def spmv(self, x, y=None):
        """SparseMatrix::spmv(&self, x, y): y <- A x.  Device vectors stay on the device; host arrays round-trip
        over PCIe (operator-level drop-in, plumbing only)."""
        if isinstance(x, DeviceVec):
            if y is None:
                y = DeviceVec(self.ctx, self._nrows)
            check(lib().kryst_spmv(self.h, x.h, y.h))
            return y
        xa = _f64(x)
        out = np.empty(self._nrows) if y is None else y
        if len(out) != self._nrows:
            raise KError(102, "spmv: y.len() != nrows")
        tmp = out if (out.dtype == np.float64 and out.flags.c_contiguous) else np.empty(self._nrows)
        check(lib().kryst_spmv_host(self.h, _dp(xa), len(xa), _dp(tmp), len(tmp)))
        if tmp is not out:
            out[:] = tmp
        return out

    ENCODINGS = ("csr", "csr-d8", "csr-d16", "csr-p16", "csr-dia")

    def encoding(self):
        """(name, patterns, table_entries) of the storage form kryst_spmv streams (see kryst_csr_encoding)."""
        e, p, t = C.c_int32(0), C.c_int32(0), C.c_int32(0)
        check(lib().kryst_csr_encoding(self.h, C.byref(e), C.byref(p), C.byref(t)))
        return self.ENCODINGS[e.value], p.value, t.value

    def tile_order(self):
        """Measurement hook (kryst_csr_tile_order): {"plane_rows", "slots", "slots8", "in_use"} of the slab order of the tiles."""
        info = (C.c_int64 * 4)()
        check(lib().kryst_csr_tile_order(self.h, info))
        return {"plane_rows": info[0], "slots": info[1], "slots8": info[2], "in_use": bool(info[3])}

    def pattern_info(self):
        """Measurement hook (kryst_csr_pattern_info): {"line", "uniform_far", "interior_first", "staged"} of the CSR-P16 form."""
        info = (C.c_int64 * 4)()
        check(lib().kryst_csr_pattern_info(self.h, info))
        return {"line": info[0], "uniform_far": bool(info[1]), "interior_first": info[2], "staged": bool(info[3])}

    def placement_info(self):
        """Where the CSR arrays live (kryst_csr_placement_info): {"tries", "chosen", "skeleton_ms": [...]} of the homes tried at creation."""
        t, c = C.c_int32(0), C.c_int32(0)
        ms = (C.c_double * 8)()
        check(lib().kryst_csr_placement_info(self.h, C.byref(t), C.byref(c), ms))
        return {"tries": t.value, "chosen": c.value, "skeleton_ms": [ms[k] for k in range(t.value)]}

    def bench_spmv(self, x, y, fused_dots=1, reps=50):
        """Average milliseconds per launch of the SpMV kernel (HIP events on the compute stream)."""
        ms = C.c_double()
        check(lib().kryst_bench_spmv(self.h, x.h, y.h, fused_dots, reps, C.byref(ms)))
        return ms.value

    def bench_spmv_fused(self, x, y, reps=20):
        """Average milliseconds per launch of the fused direction + SpMV kernel of CG / PCG (kryst_bench_spmv_fused); None when the operator
        cannot take that form."""
        ms = C.c_double()
        rc = lib().kryst_bench_spmv_fused(self.h, x.h, y.h, reps, C.byref(ms))
        if rc == 6:
            return None
        check(rc)
        return ms.value

    def halo_mode(self, mode):
        """'rccl' | 'peer' | 'query': how this row-partitioned operator's halo exchange travels (kryst_csr_halo_mode; collective except
        'query').  Returns the mode in use: 'peer' falls back to 'rccl' on every rank when a landing buffer cannot be exported / mapped or
        the test exchange does not arrive intact (KRYST_UNSUPPORTED).  A new operator starts on 'peer' when that works everywhere
        (KRYST_HALO_MODE=rccl: never)."""
        active = C.c_int32(0)
        rc = lib().kryst_csr_halo_mode(self.h, {"rccl": 0, "peer": 1, "query": -1}[mode], C.byref(active))
        if rc not in (0, 6):                     # (6 = KRYST_UNSUPPORTED: the documented fallback)
            check(rc)
        return "peer" if active.value == 1 else "rccl"

    def bench_csr_skeleton(self, x, y, reps=10):
        """Average milliseconds per launch of the plain-CSR kernel's traffic skeleton (the CSR arrays streamed, x read, y written -- no
        arithmetic; y receives garbage)."""
        ms = C.c_double()
        check(lib().kryst_bench_csr_skeleton(self.h, x.h, y.h, reps, C.byref(ms)))
        return ms.value

    matvec = spmv                                # MatVec::matvec (core/traits.rs:4-7)
    spmv_parallel = spmv                         # sparse.rs:103-114 (same arithmetic)

    def download(self):
        rp = np.empty(self._nrows + 1, dtype=np.int64)
        ci = np.empty(self.nnz, dtype=np.int32)
        va = np.empty(self.nnz)
        check(lib().kryst_csr_download(self.h, rp.ctypes.data_as(_ffi.c_i64p), ci.ctypes.data_as(_ffi.c_i32p), _dp(va)))
        return rp, ci, va

    def __del__(self):
        try:
            if self.h and self.ctx.h:
                lib().kryst_csr_destroy(self.h)
        except Exception:
            pass


# ----------------------------------------------------------------------------- preconditioners
class _Pc:
    """Preconditioner<M, V> (src/preconditioner/mod.rs:8-13): setup(&mut self, a), apply(&self, r, z)."""

    def __init__(self):
        self.h, self.ctx = None, None

    def _set(self, ctx, handle):
        self._free()
        self.ctx, self.h = ctx, handle

    def apply(self, r, z=None):
        if self.h is None:
            raise KError(2, "preconditioner used before setup")
        if isinstance(r, DeviceVec):
            z = z if z is not None else DeviceVec(self.ctx, len(r))
            check(lib().kryst_pc_apply(self.h, r.h, z.h))
            return z
        rv = DeviceVec(self.ctx, r)
        zv = DeviceVec(self.ctx, len(rv))
        check(lib().kryst_pc_apply(self.h, rv.h, zv.h))
        out = zv.to_host()
        if z is not None:
            z[:] = out
            return z
        return out

    def bench_apply(self, r, z, reps=20):
        """Average milliseconds of one apply (HIP events on the compute stream, `reps` back-to-back applies)."""
        ms = C.c_double()
        check(lib().kryst_bench_pc_apply(self.h, r.h, z.h, reps, C.byref(ms)))
        return ms.value

    def ilu_info(self):
        """What an ILU-family apply runs and streams (kryst_pc_ilu_info) -> dict."""
        v = np.zeros(13, dtype=np.int64)
        check(lib().kryst_pc_ilu_info(self.h, v.ctypes.data_as(_ffi.c_i64p), 13))
        form = ("level-ordered", "grid 8x8 (tri_wave_kernel)", "grid 16x16 (tri_quad_kernel)", "grid planes (tri_plane_kernel)",
                "box planes (tri_box_plane_kernel)", "box wavefront (tri_box_kernel)")[int(v[0])]
        if form.startswith("box"):          # info[6..8]: coefficient streams L / U have (of 13 each), both factors "regular" (tri_box.h)
            return {"form": form, "box": [int(v[1]), int(v[2]), int(v[3])], "levels": [int(v[4]), int(v[5])], "streams": [int(v[6]), int(v[7])],
                    "regular": bool(v[8]), "chunks": [0, 0], "chunks_not_requested": [0, 0], "bytes_per_chunk": [0, 0]}
        return {"form": form, "box": [int(v[1]), int(v[2]), int(v[3])], "levels": [int(v[4]), int(v[5])], "chunks": [int(v[6]), int(v[7])],
                "chunks_not_requested": [int(v[8]), int(v[9])], "bytes_per_chunk": [int(v[10]), int(v[11])]}

    def _free(self):
        try:
            if self.h and self.ctx and self.ctx.h:
                lib().kryst_pc_destroy(self.h)
        except Exception:
            pass
        self.h = None

    def __del__(self):
        self._free()


class Jacobi(_Pc):
    """Jacobi::new(); setup extracts 1/diag (src/preconditioner/jacobi.rs:26-95)."""

    def setup(self, a):
        h = _ffi.Handle()
        check(lib().kryst_pc_jacobi(a.h, C.byref(h)))
        self._set(a.ctx, h)
        self._a = a
        return self


class _IluBase(_Pc):
    MODE = 0

    def setup(self, a):
        h = _ffi.Handle()
        check(lib().kryst_pc_ilu0(a.h, self.MODE, C.byref(h)))
        self._set(a.ctx, h)
        self._a = a
        return self


class Ilu0(_IluBase):
    """Ilu0 exactly as written in src/preconditioner/ilu.rs:59-122 (L = I + tril(A,-1)D^-1, U = I + triu(A,1))."""
    MODE = 0


class Ilup(_IluBase):
    """Ilup::new(fill) exactly as written in src/preconditioner/ilup.rs:54-167 (level-of-fill p; p = 0 performs no
    elimination at all)."""
    MODE = 1

    def __init__(self, fill=0):
        super().__init__()
        self.fill = fill

    def setup(self, a):
        h = _ffi.Handle()
        check(lib().kryst_pc_ilup(a.h, self.fill, C.byref(h)))
        self._set(a.ctx, h)
        self._a = a
        return self


class Ilut(_IluBase):
    """Ilut::new(fill, droptol) exactly as written in src/preconditioner/ilut.rs:55-150 (no elimination: drop by
    magnitude, keep the `fill` largest entries of each row, split at the diagonal)."""

    def __init__(self, fill, droptol):
        super().__init__()
        self.fill, self.droptol = fill, droptol

    def setup(self, a):
        h = _ffi.Handle()
        check(lib().kryst_pc_ilut(a.h, self.fill, self.droptol, C.byref(h)))
        self._set(a.ctx, h)
        self._a = a
        return self


class TrueIlu0(_IluBase):
    """Extension: textbook ILU(0) on A's pattern (not in the reference)."""
    MODE = 2


class Chebyshev(_Pc):
    """Chebyshev::new(degree, lambda_min, lambda_max); the trait apply is a stub that returns Err
    (src/preconditioner/chebyshev.rs:35-70) -- use apply_chebyshev."""

    def __init__(self, degree, lambda_min=None, lambda_max=None):
        super().__init__()
        self.degree, self.lambda_min, self.lambda_max = degree, lambda_min, lambda_max

    def setup(self, a):
        h = _ffi.Handle()
        check(lib().kryst_pc_chebyshev_stub(a.ctx.h, self.degree, C.byref(h)))
        self._set(a.ctx, h)
        return self


class ChebyshevPc(_Pc):
    """Extension: a Preconditioner whose apply is apply_chebyshev(a, r, z, alpha, beta, degree)."""

    def __init__(self, degree, alpha, beta):
        super().__init__()
        self.degree, self.alpha, self.beta = degree, alpha, beta

    def setup(self, a):
        h = _ffi.Handle()
        check(lib().kryst_pc_chebyshev(a.h, self.alpha, self.beta, self.degree, C.byref(h)))
        self._set(a.ctx, h)
        self._a = a
        return self


class IdentityPc(_Pc):
    """The reference tests' IdentityPC (src/solver/pcg.rs:245-251)."""

    def setup(self, a):
        h = _ffi.Handle()
        check(lib().kryst_pc_identity(a.ctx.h, C.byref(h)))
        self._set(a.ctx, h)
        return self


class ApproxInv(_Pc):
    """ApproxInv (SPAI) with given inverse rows: `inv_rows[i]` = [(col, value), ...] in ascending column order, the layout of
    ApproxInv::inv_rows (src/preconditioner/approxinv.rs:66).  apply (approxinv.rs:268-298) is z = M r on the device.
    ApproxInv::setup (per-column least squares through faer's QR, approxinv.rs:129-264) stays with the reference: compute the
    rows there and hand them over; setup() here only checks the size against the operator."""

    def __init__(self, inv_rows, ctx=None):
        super().__init__()
        n = len(inv_rows)
        rp = np.zeros(n + 1, dtype=np.int64)
        for i, row in enumerate(inv_rows):
            rp[i + 1] = rp[i] + len(row)
        ci = np.array([c for row in inv_rows for c, _ in row], dtype=np.int64)
        va = np.array([v for row in inv_rows for _, v in row], dtype=np.float64)
        self.m = CsrMatrix.from_csr(n, n, rp, ci, va, ctx=ctx)
        h = _ffi.Handle()
        check(lib().kryst_pc_approx_inverse(self.m.h, C.byref(h)))
        self._set(self.m.ctx, h)

    def setup(self, a):
        if a.nrows() != self.m.nrows():
            raise KError(102, "ApproxInv: inverse rows and operator differ in size")
        return self


def apply_chebyshev(a, r, z, alpha, beta, m):
    """apply_chebyshev(a, r, z, alpha, beta, m)  src/preconditioner/chebyshev.rs:83-140."""
    if isinstance(r, DeviceVec):
        check(lib().kryst_apply_chebyshev(a.h, r.h, z.h, alpha, beta, m))
        return z
    rv = DeviceVec(a.ctx, r)
    zv = DeviceVec(a.ctx, len(rv))
    check(lib().kryst_apply_chebyshev(a.h, rv.h, zv.h, alpha, beta, m))
    z[:] = zv.to_host()
    return z


# ----------------------------------------------------------------------------- solvers
class Convergence:
    """Convergence { tol, max_iters }  src/utils/convergence.rs:4-7."""

    def __init__(self, tol, max_iters):
        self.tol, self.max_iters = tol, max_iters


class SolveStats:
    """SolveStats { iterations, final_residual, converged }  src/utils/convergence.rs:10-14."""

    def __init__(self, iterations, final_residual, converged):
        self.iterations, self.final_residual, self.converged = iterations, final_residual, converged

    def __repr__(self):
        return (f"SolveStats {{ iterations: {self.iterations}, final_residual: {self.final_residual:e}, "
                f"converged: {self.converged} }}")


class CgNormType(enum.IntEnum):                  # src/solver/cg.rs:35
    Preconditioned = 0
    Unpreconditioned = 1
    Natural = 2
    NoNorm = 3


class Preconditioning(enum.IntEnum):             # src/solver/gmres.rs:28-32
    NoPc = 0
    Left = 1
    Right = 2
    LeftTextbook = 3                             # LABELLED EXTENSION (not in the reference): Arnoldi on M^-1 A from M^-1 r0, Gram-Schmidt against V --
                                                 # the reference's Left orthogonalises against an un-normalised Z[0] (gmres.rs:240-247, 279-307)


class _Solver:
    _HOST = _DEV = None
    _HIST_PER_ITER = 1

    def __init__(self, tol, max_iters):
        self.conv = Convergence(tol, max_iters)
        self.norm_type = CgNormType.Unpreconditioned
        self.single_reduction = False
        self.radius = None
        self.obj_target = None
        self.monitor = None
        self.residual_history = []
        self.restart = 0
        self.preconditioning = Preconditioning.Left
        self.check_every = 0

    def _params(self):
        return _ffi.Params(self.conv.tol, self.conv.max_iters, self.restart, int(self.preconditioning),
                           int(self.norm_type), int(self.single_reduction),
                           int(self.radius is not None), self.radius or 0.0,
                           int(self.obj_target is not None), self.obj_target or 0.0, self.check_every)

    def solve(self, a, pc, b, x):
        """LinearSolver::solve(&mut self, a, pc: Option<&dyn Preconditioner>, b, x) -> Result<SolveStats, KError>
        (src/solver/mod.rs:43-49).  x is in/out.  Host arrays are uploaded / downloaded around the device solve;
        DeviceVec arguments stay in HBM."""
        prm = self._params()
        st = _ffi.Stats()
        cap = min(self._HIST_PER_ITER * self.conv.max_iters + max(self.restart, 1) + 8, (1 << 22) + 8)   # the library records at most 2^22 entries
        hist = np.zeros(cap)
        hlen = C.c_int64(0)
        cb = _ffi.MONITOR(lambda it, res, _u: self.monitor(it, res)) if self.monitor else _ffi.MONITOR()
        pch = pc.h if pc is not None else None
        if pc is not None and pch is None:
            raise KError(2, "preconditioner used before setup")
        tail = self._extra() + (a.h, pch, C.byref(prm), C.byref(st), _dp(hist), cap, C.byref(hlen), cb, None)
        if isinstance(b, DeviceVec):
            rc = getattr(lib(), self._DEV)(b.h, x.h, *tail)
        elif self._HOST is None:                     # extension solvers only exist in device-vector form
            bv, xv = DeviceVec(a.ctx, b), DeviceVec(a.ctx, x)
            rc = getattr(lib(), self._DEV)(bv.h, xv.h, *tail)
            if rc == 0:
                x[:] = xv.to_host()
        else:
            bb = _f64(b)
            if not (isinstance(x, np.ndarray) and x.dtype == np.float64 and x.flags.c_contiguous):
                raise KError(102, "x must be a contiguous float64 numpy array (it is written in place)")
            if len(bb) != len(x):
                raise KError(102, "b and x differ in length")
            rc = getattr(lib(), self._HOST)(_dp(bb), _dp(x), len(bb), *tail)
        self.residual_history.extend(hist[:min(hlen.value, cap)].tolist())
        stats = SolveStats(st.iterations, st.final_residual, bool(st.converged))
        check(rc, stats)
        return stats

    def _extra(self):
        return ()

    def clear_history(self):
        self.residual_history.clear()

    # builder methods (cg.rs:64-88)
    def with_norm(self, norm_type):
        self.norm_type = norm_type
        return self

    def with_single_reduction(self, flag):
        self.single_reduction = flag
        return self

    def with_radius(self, radius):
        self.radius = radius
        return self

    def with_obj_target(self, obj):
        self.obj_target = obj
        return self

    def with_monitor(self, f):
        self.monitor = f
        return self


class CgSolver(_Solver):
    """CgSolver::new(tol, max_iters)  src/solver/cg.rs:40-93,114-288 (pc is ignored, cg.rs:115)."""
    _HOST, _DEV = "kryst_cg_solve", "kryst_cg_solve_dev"


class PcgSolver(_Solver):
    """PcgSolver::new(tol, max_iters)  src/solver/pcg.rs:31-91,114-222."""
    _HOST, _DEV = "kryst_pcg_solve", "kryst_pcg_solve_dev"


class GmresSolver(_Solver):
    """GmresSolver::new(restart, tol, max_iters)  src/solver/gmres.rs:38-60,216-402."""
    _HOST, _DEV = "kryst_gmres_solve", "kryst_gmres_solve_dev"

    def __init__(self, restart, tol, max_iters):
        super().__init__(tol, max_iters)
        self.restart = restart

    def with_preconditioning(self, mode):
        self.preconditioning = mode
        return self


class Orthog(enum.IntEnum):                      # src/solver/fgmres.rs:26-31
    Classical = 0
    Modified = 1


class FgmresSolver(_Solver):
    """FgmresSolver::new(tol, max_iters, restart)  src/solver/fgmres.rs:33-101; solve_flex :114-340.  The flexible
    preconditioner is any device preconditioner object (its action is fixed per apply, which FGMRES permits);
    `solve` forwards to solve_flex so the solver also fits the LinearSolver call shape used elsewhere."""
    _HOST, _DEV = "kryst_fgmres_solve", "kryst_fgmres_solve_dev"

    def __init__(self, tol, max_iters, restart):
        super().__init__(tol, max_iters)
        self.restart = restart
        self.orthog = Orthog.Classical
        self.haptol = 1e-12
        self.preallocate = False
        self.delta_allocate = 10                     # accepted, no effect (allocation granularity, fgmres.rs:77-80)

    def _extra(self):
        return (int(self.orthog), float(self.haptol), int(self.preallocate))

    def with_orthog(self, orthog):
        self.orthog = orthog
        return self

    def with_preallocate(self, flag):
        self.preallocate = flag
        return self

    def with_delta_allocate(self, delta):
        self.delta_allocate = delta
        return self

    def with_haptol(self, haptol):
        self.haptol = haptol
        return self

    def solve_flex(self, a, pc, b, x):
        return self.solve(a, pc, b, x)


class BiCgStabSolver(_Solver):
    """BiCgStabSolver::new(tol, max_iters)  src/solver/bicgstab.rs:36-48,69-293 (pc ignored, absolute tolerance)."""
    _HOST, _DEV = "kryst_bicgstab_solve", "kryst_bicgstab_solve_dev"


class CgsSolver(_Solver):
    """CgsSolver::new(tol, max_iters)  src/solver/cgs.rs:21-35,58-135 (pc ignored, :59)."""
    _HOST, _DEV = "kryst_cgs_solve", "kryst_cgs_solve_dev"


class TfqmrSolver(_Solver):
    """TfqmrSolver::new(tol, max_iters)  src/solver/tfqmr.rs:30-40,64-221 as written (pc ignored, x starts from zero whatever
    the caller passes, :72).  residual_history receives the residual estimate of both substeps."""
    _HOST, _DEV = "kryst_tfqmr_solve", "kryst_tfqmr_solve_dev"
    _HIST_PER_ITER = 2


class BiCgStabRightPcSolver(_Solver):
    """Extension: right-preconditioned BiCGStab (device vectors only)."""
    _HOST, _DEV = None, "kryst_bicgstab_rpc_solve_dev"


class PC:
    """PC<T> (src/context/pc_context.rs:36-76): the reference's configuration enum for preconditioners, plus the constructor it
    lacks -- `PC.Ilut(fill=10, droptol=1e-3).build(a)` returns the set-up device preconditioner.  Kinds outside the hot path
    (Ssor, ApproxInv setup, BlockJacobi, Multicolor, AMG, AdditiveSchwarz) raise KError(Unsupported)."""

    def __init__(self, kind, **params):
        self.kind, self.params = kind, params

    def __repr__(self):
        return f"PC::{self.kind}{self.params or ''}"

    @staticmethod
    def Jacobi():
        return PC("Jacobi")

    @staticmethod
    def Ilu0():
        return PC("Ilu0")

    @staticmethod
    def Ilup(fill):
        return PC("Ilup", fill=fill)

    @staticmethod
    def Ilut(fill, droptol):
        return PC("Ilut", fill=fill, droptol=droptol)

    @staticmethod
    def Chebyshev(degree, emin=None, emax=None):
        return PC("Chebyshev", degree=degree, emin=emin, emax=emax)

    def build(self, a):
        k, q = self.kind, self.params
        if k == "Jacobi":
            return Jacobi().setup(a)
        if k == "Ilu0":
            return Ilu0().setup(a)
        if k == "Ilup":
            return Ilup(q["fill"]).setup(a)
        if k == "Ilut":
            return Ilut(q["fill"], q["droptol"]).setup(a)
        if k == "Chebyshev":                         # the trait object of the reference (apply is the stub of chebyshev.rs:68-70)
            return Chebyshev(q["degree"], q["emin"], q["emax"]).setup(a)
        raise KError(6, f"preconditioner kind {k} is outside the accelerated path")


class SolverKind(enum.Enum):                      # src/context/ksp_context.rs:25-48 (the kinds on the hot path)
    Cg = "cg"
    Pcg = "pcg"
    GmresLeft = "gmres_left"
    GmresRight = "gmres_right"
    Bicgstab = "bicgstab"
    Fgmres = "fgmres"
    Cgs = "cgs"
    Tfqmr = "tfqmr"


class KspContext:
    """KspContext { kind, a, pc, tol, max_it, restart } + solve_context (src/context/ksp_context.rs:54-148): builds a
    fresh solver per call and forwards (a, pc, b, x).  Kinds outside the hot path raise KError(Unsupported)."""

    def __init__(self, kind, a, pc=None, tol=1e-8, max_it=1000, restart=30, flex_pc=None):
        self.kind, self.a, self.pc, self.tol, self.max_it, self.restart = kind, a, pc, tol, max_it, restart
        self.flex_pc = flex_pc                      # ksp_context.rs:62: FGMRES uses flex_pc, never pc (:101-107)

    def solve_context(self, b, x, comm=None):
        k = self.kind
        if k == SolverKind.GmresLeft:
            s = GmresSolver(self.restart, self.tol, self.max_it).with_preconditioning(Preconditioning.Left)
        elif k == SolverKind.GmresRight:
            s = GmresSolver(self.restart, self.tol, self.max_it).with_preconditioning(Preconditioning.Right)
        elif k == SolverKind.Cg:
            s = CgSolver(self.tol, self.max_it)
        elif k == SolverKind.Pcg:
            s = PcgSolver(self.tol, self.max_it)
        elif k == SolverKind.Bicgstab:
            s = BiCgStabSolver(self.tol, self.max_it)
        elif k == SolverKind.Cgs:
            s = CgsSolver(self.tol, self.max_it)
        elif k == SolverKind.Tfqmr:
            s = TfqmrSolver(self.tol, self.max_it)
        elif k == SolverKind.Fgmres:
            return FgmresSolver(self.tol, self.max_it, self.restart).solve_flex(self.a, self.flex_pc, b, x)
        else:
            raise KError(6, f"solver kind {k} is outside the accelerated path")
        return s.solve(self.a, self.pc, b, x)


class Session:
    """Stepping form of CgSolver / PcgSolver / BiCgStabSolver on device vectors: begin, step(k) (enqueue k
    iterations without synchronising), end() -> SolveStats.  bench.py uses it to time exactly K iterations."""
    METHODS = {"cg": 0, "pcg": 1, "bicgstab": 2, "cgs": 3, "tfqmr": 4}

    def __init__(self, method, a, pc, b, x, tol, max_iters, norm_type=CgNormType.Unpreconditioned):
        self.a, self.pc, self.b, self.x = a, pc, b, x
        self.max_iters = max_iters
        prm = _ffi.Params(tol, max_iters, 0, 1, int(norm_type), 0, 0, 0.0, 0, 0.0, 0)
        self.h = _ffi.Handle()
        check(lib().kryst_session_begin(self.METHODS[method], b.h, x.h, a.h, pc.h if pc is not None else None,
                                        C.byref(prm), C.byref(self.h)))

    def step(self, k):
        check(lib().kryst_session_step(self.h, k))

    # A session that is never ended keeps its context busy (KRYST_ERR_BUSY for every later solve): `with Session(...) as s:`
    # and the finaliser end it on every path (an exception between begin and end, a failing step)
    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

    def close(self):
        if getattr(self, "h", None):
            st = _ffi.Stats()
            hlen = C.c_int64(0)
            try:
                lib().kryst_session_end(self.h, C.byref(st), None, 0, C.byref(hlen))
            finally:
                self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def end(self):
        st = _ffi.Stats()
        cap = self.max_iters + 8
        hist = np.zeros(cap)
        hlen = C.c_int64(0)
        rc = lib().kryst_session_end(self.h, C.byref(st), _dp(hist), cap, C.byref(hlen))
        self.h = None
        self.residual_history = hist[:min(hlen.value, cap)].tolist()
        stats = SolveStats(st.iterations, st.final_residual, bool(st.converged))
        check(rc, stats)
        return stats


# ----------------------------------------------------------------------------- host-only helpers
def _read_matrix_file(fn, path):
    nr, nc = C.c_int64(0), C.c_int64(0)
    nnz = fn(str(path).encode(), C.byref(nr), C.byref(nc), None, None, None)
    if nnz < 0:
        raise KError(102, lib().kryst_hip_last_error().decode())
    rp = np.zeros(nr.value + 1, dtype=np.int64); ci = np.zeros(max(nnz, 1), dtype=np.int64); va = np.zeros(max(nnz, 1))
    got = fn(str(path).encode(), C.byref(nr), C.byref(nc), rp.ctypes.data_as(_ffi.c_i64p), ci.ctypes.data_as(_ffi.c_i64p), _dp(va))
    if got != nnz:
        raise KError(102, lib().kryst_hip_last_error().decode() if got < 0 else "matrix file changed while reading")
    return nr.value, nc.value, rp, ci[:nnz], va[:nnz]


def read_matrix_market(path):
    """Matrix Market coordinate file -> (nrows, ncols, row_ptr, col_idx, vals) (host only; kryst_host_read_matrix_market)."""
    return _read_matrix_file(lib().kryst_host_read_matrix_market, path)


def read_petsc_binary(path):
    """PETSc binary AIJ matrix -> (nrows, ncols, row_ptr, col_idx, vals) (host only; kryst_host_read_petsc_binary)."""
    return _read_matrix_file(lib().kryst_host_read_petsc_binary, path)


def _host_factors(h):
    try:
        n, nl, nu = C.c_int64(0), C.c_int64(0), C.c_int64(0)
        check(lib().kryst_host_factors_sizes(h, C.byref(n), C.byref(nl), C.byref(nu)))
        lp = np.zeros(n.value + 1, dtype=np.int64); up = np.zeros(n.value + 1, dtype=np.int64)
        lc = np.zeros(max(nl.value, 1), dtype=np.int32); uc = np.zeros(max(nu.value, 1), dtype=np.int32)
        lv = np.zeros(max(nl.value, 1)); uv = np.zeros(max(nu.value, 1)); dg = np.zeros(max(n.value, 1))
        check(lib().kryst_host_factors_get(h, lp.ctypes.data_as(_ffi.c_i64p), lc.ctypes.data_as(_ffi.c_i32p), _dp(lv),
                                           up.ctypes.data_as(_ffi.c_i64p), uc.ctypes.data_as(_ffi.c_i32p), _dp(uv), _dp(dg)))
        return lp, lc[:nl.value], lv[:nl.value], up, uc[:nu.value], uv[:nu.value], dg[:n.value]
    finally:
        lib().kryst_host_factors_destroy(h)


def _host_rows(row_ptr, col_idx, values):
    rp = np.ascontiguousarray(row_ptr, dtype=np.int64)
    ci = np.ascontiguousarray(col_idx, dtype=np.int32)
    va = _f64(values)
    if len(rp) < 1 or len(ci) != len(va) or int(rp[-1]) != len(va):
        raise KError(102, "host factorisation: inconsistent array lengths")
    return rp, ci, va


def host_ilup(row_ptr, col_idx, values, fill, threads=0, block=0):
    """Ilup::new(fill).setup (ilup.rs:77-134) on host arrays, no GPU (kryst_host_ilup): the row pipeline kryst_pc_ilup runs between download
    and upload.  -> (l_ptr, l_col, l_val, u_ptr, u_col, u_val, diag): L's strictly-lower multipliers, U's strictly-upper kept entries, the kept
    diagonal (1.0 where none is kept).  KError(SolveError) on a zero u_jj, `.row` = that j."""
    rp, ci, va = _host_rows(row_ptr, col_idx, values)
    h = _ffi.Handle()
    check(lib().kryst_host_ilup(len(rp) - 1, rp.ctypes.data_as(_ffi.c_i64p), ci.ctypes.data_as(_ffi.c_i32p), _dp(va), fill, threads, block, C.byref(h)))
    return _host_factors(h)


def host_ilut(row_ptr, col_idx, values, fill, droptol, threads=0):
    """Ilut::new(fill, droptol).setup (ilut.rs:80-117) on host arrays, no GPU (kryst_host_ilut); result as host_ilup."""
    rp, ci, va = _host_rows(row_ptr, col_idx, values)
    h = _ffi.Handle()
    check(lib().kryst_host_ilut(len(rp) - 1, rp.ctypes.data_as(_ffi.c_i64p), ci.ctypes.data_as(_ffi.c_i32p), _dp(va), fill, float(droptol), threads, C.byref(h)))
    return _host_factors(h)


def host_levels(ptr, col, forward=True):
    """Dependency levels of a strictly-lower (forward) / strictly-upper triangular factor's rows (kryst_host_levels) -> (level[n], nlevels)."""
    p = np.ascontiguousarray(ptr, dtype=np.int64)
    c = np.ascontiguousarray(col, dtype=np.int32)
    lvl = np.zeros(max(len(p) - 1, 1), dtype=np.int32)
    nl = C.c_int32(0)
    check(lib().kryst_host_levels(len(p) - 1, p.ctypes.data_as(_ffi.c_i64p), c.ctypes.data_as(_ffi.c_i32p), 1 if forward else 0,
                                  lvl.ctypes.data_as(_ffi.c_i32p), C.byref(nl)))
    return lvl[:len(p) - 1], nl.value


def host_stencil7(N, kind="poisson", k_lo=0, k_hi=None):
    k_hi = N if k_hi is None else k_hi
    kk = STENCIL_KINDS[kind]
    nnz = lib().kryst_host_stencil7(N, kk, k_lo, k_hi, None, None, None)
    if nnz < 0:
        raise KError(102, "host_stencil7")
    nloc = (k_hi - k_lo) * N * N
    rp = np.empty(nloc + 1, dtype=np.int64); ci = np.empty(nnz, dtype=np.int64); va = np.empty(nnz)
    lib().kryst_host_stencil7(N, kk, k_lo, k_hi, rp.ctypes.data_as(_ffi.c_i64p), ci.ctypes.data_as(_ffi.c_i64p), _dp(va))
    return rp, ci, va


def partition_rows(n, nranks, align=1):
    out = np.empty(nranks + 1, dtype=np.int64)
    check(lib().kryst_host_partition_rows(n, nranks, align, out.ctypes.data_as(_ffi.c_i64p)))
    return out


def halo_recv_plan(rank, nranks, row_offsets, row_ptr, col_idx_global):
    ro = np.ascontiguousarray(row_offsets, dtype=np.int64)
    rp = np.ascontiguousarray(row_ptr, dtype=np.int64)
    ci = np.ascontiguousarray(col_idx_global, dtype=np.int64)
    args = (rank, nranks, ro.ctypes.data_as(_ffi.c_i64p), rp.ctypes.data_as(_ffi.c_i64p), ci.ctypes.data_as(_ffi.c_i64p))
    total = lib().kryst_host_halo_recv_plan(*args, None, None)
    counts = np.zeros(nranks, dtype=np.int64); cols = np.zeros(max(total, 1), dtype=np.int64)
    lib().kryst_host_halo_recv_plan(*args, counts.ctypes.data_as(_ffi.c_i64p), cols.ctypes.data_as(_ffi.c_i64p))
    return counts, cols[:total]
