"""ctypes front-end of the CPU oracle (oracle/kryst_oracle.c).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the product
package kryst_amd never does.  See oracle/kryst_oracle.h for the contract and the reference citations.

Also holds numpy problem generators used by the tests (SURVEY.md section 8d definitions).  They are written
independently of the product's generator (kryst_amd/csrc/problems.cpp) so that each checks the other.
"""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "libkryst_oracle.so")

OK, FACTOR_ERROR, SOLVE_ERROR, INDEFINITE_MATRIX, INDEFINITE_PC, ZERO_PIVOT, UNSUPPORTED = range(7)
REDUCE_SERIAL, REDUCE_TILED = 0, 1
PC_NONE, PC_IDENTITY, PC_JACOBI, PC_ILU0_COMPAT, PC_ILUP0, PC_ILU0_TRUE, PC_CHEB_STUB, PC_CHEB, PC_TRIROWS = range(9)
SIDE_NONE, SIDE_LEFT, SIDE_RIGHT = 0, 1, 2
SIDE_LEFT_TEXTBOOK = 3          # labelled extension (kro_gmres side 3): Arnoldi on M^-1 A from M^-1 r0, Gram-Schmidt against V
NORM_PRECONDITIONED, NORM_UNPRECONDITIONED, NORM_NATURAL, NORM_NONE = 0, 1, 2, 3

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int64)


class _Reduce(C.Structure):
    _fields_ = [("mode", C.c_int32), ("T", C.c_int32), ("V", C.c_int32), ("F", C.c_int32),
                ("nparts", C.c_int32), ("part_off", _ip)]


class _Csr(C.Structure):
    _fields_ = [("nrows", C.c_int64), ("ncols", C.c_int64), ("row_ptr", _ip), ("col_idx", _ip), ("vals", _dp)]


class _TriRows(C.Structure):
    _fields_ = [("n", C.c_int64), ("l_ptr", _ip), ("l_col", _ip), ("l_val", _dp), ("u_ptr", _ip), ("u_col", _ip), ("u_val", _dp)]


class _Pc(C.Structure):
    _fields_ = [("kind", C.c_int32), ("a", C.POINTER(_Csr)), ("inv_diag", _dp), ("lfac", _dp), ("ufac", _dp),
                ("divide_diag", C.c_int32), ("cheb_alpha", C.c_double), ("cheb_beta", C.c_double),
                ("cheb_degree", C.c_int32), ("rows", C.POINTER(_TriRows))]


class _Params(C.Structure):
    _fields_ = [("tol", C.c_double), ("max_iters", C.c_int64), ("restart", C.c_int32),
                ("precond_side", C.c_int32), ("norm_type", C.c_int32), ("single_reduction", C.c_int32),
                ("has_radius", C.c_int32), ("radius", C.c_double),
                ("has_obj_target", C.c_int32), ("obj_target", C.c_double)]


class _Stats(C.Structure):
    _fields_ = [("iterations", C.c_int64), ("final_residual", C.c_double), ("converged", C.c_int32)]


_MONITOR = C.CFUNCTYPE(None, C.c_int64, C.c_double, C.c_void_p)


class _Trace(C.Structure):
    _fields_ = [("hist", _dp), ("cap", C.c_int64), ("len", C.c_int64), ("monitor", _MONITOR), ("user", C.c_void_p)]


def build(force=False):
    """Compile the oracle with gcc (oracle/Makefile)."""
    if force or not os.path.exists(_LIB_PATH) or \
            os.path.getmtime(_LIB_PATH) < os.path.getmtime(os.path.join(_HERE, "kryst_oracle.c")):
        subprocess.run(["make", "-C", _HERE], check=True, capture_output=True)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        L.kro_dot.restype = C.c_double
        L.kro_dot.argtypes = [C.POINTER(_Reduce), _dp, _dp, C.c_int64]
        L.kro_norm.restype = C.c_double
        L.kro_norm.argtypes = [C.POINTER(_Reduce), _dp, C.c_int64]
        L.kro_spmv.restype = None
        L.kro_spmv.argtypes = [C.POINTER(_Csr), _dp, _dp]
        L.kro_csr_check.restype = C.c_int32
        L.kro_csr_check.argtypes = [C.POINTER(_Csr)]
        for nm in ("kro_jacobi_setup",):
            getattr(L, nm).restype = C.c_int32
            getattr(L, nm).argtypes = [C.POINTER(_Csr), _dp]
        for nm in ("kro_ilu0_compat_setup", "kro_ilup0_setup", "kro_ilu0_true_setup"):
            getattr(L, nm).restype = C.c_int32
            getattr(L, nm).argtypes = [C.POINTER(_Csr), _dp, _dp]
        L.kro_ilup_build.restype = C.c_int32
        L.kro_ilup_build.argtypes = [C.POINTER(_Csr), C.c_int64, C.POINTER(_TriRows)]
        L.kro_ilut_build.restype = C.c_int32
        L.kro_ilut_build.argtypes = [C.POINTER(_Csr), C.c_int64, C.c_double, C.POINTER(_TriRows)]
        L.kro_trirows_free.argtypes = [C.POINTER(_TriRows)]
        L.kro_pc_apply.restype = C.c_int32
        L.kro_pc_apply.argtypes = [C.POINTER(_Pc), _dp, _dp, C.c_int64]
        L.kro_apply_chebyshev.restype = None
        L.kro_apply_chebyshev.argtypes = [C.POINTER(_Csr), _dp, _dp, C.c_int64, C.c_double, C.c_double, C.c_int64]
        L.kro_chebyshev_t.restype = C.c_double
        L.kro_chebyshev_t.argtypes = [C.c_int64, C.c_double]
        for nm in ("kro_cg", "kro_pcg", "kro_gmres", "kro_bicgstab", "kro_bicgstab_rpc"):
            getattr(L, nm).restype = C.c_int32
            getattr(L, nm).argtypes = [C.POINTER(_Csr), C.POINTER(_Pc), _dp, _dp, C.POINTER(_Params),
                                       C.POINTER(_Reduce), C.POINTER(_Stats), C.POINTER(_Trace)]
        for nm in ("kro_cgs", "kro_tfqmr"):
            getattr(L, nm).restype = C.c_int32
            getattr(L, nm).argtypes = [C.POINTER(_Csr), C.POINTER(_Pc), _dp, _dp, C.POINTER(_Params), C.POINTER(_Reduce),
                                       C.POINTER(_Stats), C.POINTER(_Trace)]
        L.kro_fgmres.restype = C.c_int32
        L.kro_fgmres.argtypes = [C.POINTER(_Csr), C.POINTER(_Pc), _dp, _dp, C.POINTER(_Params), C.c_int32, C.c_double, C.c_int32,
                                 C.POINTER(_Reduce), C.POINTER(_Stats), C.POINTER(_Trace)]
        L.kro_set_threads.argtypes = [C.c_int32]
        L.kro_get_threads.restype = C.c_int32
        _lib = L
    return _lib


def set_threads(n):
    lib().kro_set_threads(int(n))


def _d(a):
    return a.ctypes.data_as(_dp)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


class Reduce:
    """Inner-product association order (kryst_oracle.h)."""

    def __init__(self, mode=REDUCE_SERIAL, T=256, V=2, F=1024, part_off=None):
        self.part_off = None if part_off is None else np.ascontiguousarray(part_off, dtype=np.int64)
        self.c = _Reduce(mode, T, V, F,
                         0 if self.part_off is None else len(self.part_off) - 1,
                         None if self.part_off is None else self.part_off.ctypes.data_as(_ip))

    @staticmethod
    def serial():
        return Reduce(REDUCE_SERIAL)

    @staticmethod
    def tiled(T=256, V=2, F=1024, part_off=None):
        return Reduce(REDUCE_TILED, T, V, F, part_off)

    def ref(self):
        return C.byref(self.c)


SERIAL = Reduce.serial()


class Csr:
    """CSR in the reference's from_csr layout (sparse.rs:28-46): usize row_ptr / col_idx, f64 values."""

    def __init__(self, nrows, ncols, row_ptr, col_idx, vals, check=True):
        self.nrows, self.ncols = int(nrows), int(ncols)
        self.row_ptr = np.ascontiguousarray(row_ptr, dtype=np.int64)
        self.col_idx = np.ascontiguousarray(col_idx, dtype=np.int64)
        self.vals = _f64(vals)
        assert len(self.row_ptr) == self.nrows + 1 and len(self.col_idx) == len(self.vals) == self.row_ptr[-1]
        self.c = _Csr(self.nrows, self.ncols, self.row_ptr.ctypes.data_as(_ip),
                      self.col_idx.ctypes.data_as(_ip), _d(self.vals))
        if check:
            rc = lib().kro_csr_check(C.byref(self.c))
            if rc:
                raise ValueError(f"CSR precondition violated (code {rc})")

    @property
    def nnz(self):
        return int(self.row_ptr[-1])

    @staticmethod
    def from_dense(a, keep_zeros=True):
        """Dense matrix as CSR.  keep_zeros=True stores every entry, which reproduces the reference's dense row
        loop (wrappers.rs:31-36) term by term, signed zeros included."""
        a = np.asarray(a, dtype=np.float64)
        n, m = a.shape
        if keep_zeros:
            rp = np.arange(0, n * m + 1, m, dtype=np.int64)
            ci = np.tile(np.arange(m, dtype=np.int64), n)
            return Csr(n, m, rp, ci, a.reshape(-1).copy())
        rp = [0]; ci = []; va = []
        for i in range(n):
            for j in range(m):
                if a[i, j] != 0.0:
                    ci.append(j); va.append(a[i, j])
            rp.append(len(ci))
        return Csr(n, m, rp, ci, va)

    def to_dense(self):
        a = np.zeros((self.nrows, self.ncols))
        for i in range(self.nrows):
            for k in range(self.row_ptr[i], self.row_ptr[i + 1]):
                a[i, self.col_idx[k]] = self.vals[k]
        return a

    def spmv(self, x):
        x = _f64(x)
        assert len(x) == self.ncols
        y = np.empty(self.nrows)
        lib().kro_spmv(C.byref(self.c), _d(x), _d(y))
        return y


def dot(x, y, rs=SERIAL):
    x, y = _f64(x), _f64(y)
    assert len(x) == len(y)
    return lib().kro_dot(rs.ref(), _d(x), _d(y), len(x))


def norm(x, rs=SERIAL):
    x = _f64(x)
    return lib().kro_norm(rs.ref(), _d(x), len(x))


class Pc:
    """Preconditioner record handed to the oracle solvers."""

    def __init__(self, kind, a=None):
        self.kind, self.a = kind, a
        self.inv_diag = self.lfac = self.ufac = None
        self.c = _Pc()
        self.c.kind = kind
        if a is not None:
            self.c.a = C.pointer(a.c)

    @staticmethod
    def identity():
        return Pc(PC_IDENTITY)

    @staticmethod
    def approx_inverse(m):
        """ApproxInv::apply with the given inverse rows as the CSR matrix m (approxinv.rs:268-298)."""
        return Pc(9, m)

    @staticmethod
    def jacobi(a):
        p = Pc(PC_JACOBI, a)
        p.inv_diag = np.empty(a.nrows)
        rc = lib().kro_jacobi_setup(C.byref(a.c), _d(p.inv_diag))
        assert rc == OK
        p.c.inv_diag = _d(p.inv_diag)
        return p

    @staticmethod
    def _ilu(a, kind, fn, divide):
        p = Pc(kind, a)
        p.lfac = np.empty(a.nnz); p.ufac = np.empty(a.nnz)
        rc = getattr(lib(), fn)(C.byref(a.c), _d(p.lfac), _d(p.ufac))
        if rc:
            raise KrylovError(rc)
        p.c.lfac, p.c.ufac, p.c.divide_diag = _d(p.lfac), _d(p.ufac), divide
        return p

    @staticmethod
    def ilu0_compat(a):
        return Pc._ilu(a, PC_ILU0_COMPAT, "kro_ilu0_compat_setup", 0)

    @staticmethod
    def ilup0(a):
        return Pc._ilu(a, PC_ILUP0, "kro_ilup0_setup", 1)

    @staticmethod
    def ilu0_true(a):
        return Pc._ilu(a, PC_ILU0_TRUE, "kro_ilu0_true_setup", 1)

    @staticmethod
    def _trirows(a, rc, t):
        if rc:
            raise KrylovError(rc)
        p = Pc(PC_TRIROWS, a)
        p.tri = t
        p.c.rows = C.pointer(t)
        return p

    @staticmethod
    def ilup(a, fill):
        """Ilup::new(fill).setup(a) as written (dense work arrays: small n only)."""
        t = _TriRows()
        return Pc._trirows(a, lib().kro_ilup_build(C.byref(a.c), fill, C.byref(t)), t)

    @staticmethod
    def ilut(a, fill, droptol):
        """Ilut::new(fill, droptol).setup(a) as written."""
        t = _TriRows()
        return Pc._trirows(a, lib().kro_ilut_build(C.byref(a.c), fill, droptol, C.byref(t)), t)

    def tri_rows(self):
        """(l_ptr, l_col, l_val, u_ptr, u_col, u_val) of a TRIROWS preconditioner as numpy arrays."""
        t, n = self.tri, self.tri.n
        lp = np.ctypeslib.as_array(t.l_ptr, (n + 1,)).copy(); up = np.ctypeslib.as_array(t.u_ptr, (n + 1,)).copy()
        f = lambda ptr, m, dt: (np.ctypeslib.as_array(ptr, (max(m, 1),))[:m].copy() if m else np.zeros(0, dtype=dt))
        return lp, f(t.l_col, lp[-1], np.int64), f(t.l_val, lp[-1], float), up, f(t.u_col, up[-1], np.int64), f(t.u_val, up[-1], float)

    @staticmethod
    def chebyshev_stub():
        return Pc(PC_CHEB_STUB)

    @staticmethod
    def chebyshev(a, alpha, beta, degree):
        p = Pc(PC_CHEB, a)
        p.c.cheb_alpha, p.c.cheb_beta, p.c.cheb_degree = alpha, beta, degree
        return p

    def apply(self, r):
        r = _f64(r)
        z = np.zeros(len(r))
        rc = lib().kro_pc_apply(C.byref(self.c), _d(r), _d(z), len(r))
        if rc:
            raise KrylovError(rc)
        return z


class KrylovError(Exception):
    NAMES = {1: "FactorError", 2: "SolveError", 3: "IndefiniteMatrix", 4: "IndefinitePreconditioner",
             5: "ZeroPivot", 6: "Unsupported"}

    def __init__(self, code, stats=None):
        super().__init__(self.NAMES.get(code, f"code {code}"))
        self.code, self.stats = code, stats


def apply_chebyshev(a, r, alpha, beta, m):
    r = _f64(r)
    z = np.zeros(len(r))
    lib().kro_apply_chebyshev(C.byref(a.c), _d(r), _d(z), len(r), alpha, beta, m)
    return z


def chebyshev_t(m, x):
    return lib().kro_chebyshev_t(m, x)


class Result:
    def __init__(self, x, stats, history, code):
        self.x, self.history, self.code = x, history, code
        self.iterations, self.final_residual, self.converged = stats.iterations, stats.final_residual, bool(stats.converged)

    def __repr__(self):
        return (f"Result(code={self.code}, iterations={self.iterations}, final_residual={self.final_residual:.6e}, "
                f"converged={self.converged})")


def solve(method, a, b, x0=None, pc=None, tol=1e-8, max_iters=1000, restart=30, side=SIDE_LEFT,
          norm_type=NORM_UNPRECONDITIONED, single_reduction=False, radius=None, obj_target=None,
          rs=SERIAL, monitor=None, raise_on_error=True, orthog=0, haptol=1e-12, preallocate=False):
    """method in {"cg","pcg","gmres","bicgstab","bicgstab_rpc"}; returns Result (x, stats, residual history)."""
    b = _f64(b)
    x = np.zeros(a.nrows) if x0 is None else _f64(x0).copy()
    prm = _Params(tol, max_iters, restart, side, norm_type, int(single_reduction),
                  int(radius is not None), 0.0 if radius is None else radius,
                  int(obj_target is not None), 0.0 if obj_target is None else obj_target)
    st = _Stats()
    cap = (2 if method == "tfqmr" else 1) * max_iters + restart + 8
    hist = np.zeros(cap)
    cb = _MONITOR(lambda it, res, _u: monitor(it, res)) if monitor else _MONITOR()
    tr = _Trace(_d(hist), cap, 0, cb, None)
    fn = getattr(lib(), "kro_" + method)
    if method == "fgmres":
        rc = fn(C.byref(a.c), C.byref(pc.c) if pc is not None else None, _d(b), _d(x), C.byref(prm), int(orthog), float(haptol),
                int(preallocate), rs.ref(), C.byref(st), C.byref(tr))
    else:
        rc = fn(C.byref(a.c), C.byref(pc.c) if pc is not None else None, _d(b), _d(x), C.byref(prm), rs.ref(),
                C.byref(st), C.byref(tr))
    res = Result(x, st, hist[:min(tr.len, cap)].copy(), rc)
    if rc and raise_on_error:
        raise KrylovError(rc, res)
    return res


# --------------------------------------------------------------------------- problem generators

def stencil7(N, kind="poisson", k_lo=0, k_hi=None):
    """7-point stencil on an N^3 grid, row = i + N*(j + N*k), Dirichlet by truncation, ascending columns.
    Returns rows of planes k_lo..k_hi-1 with GLOBAL column indices.
      poisson: diag 6, off -1.   aniso: (cx,cy,cz)=(1,1,0.01).   convdiff: first-order upwind, gamma=(1,.5,.25)."""
    k_hi = N if k_hi is None else k_hi
    if kind == "poisson":
        w, e, s, nn, bo, t, dg = -1.0, -1.0, -1.0, -1.0, -1.0, -1.0, 6.0
    elif kind == "aniso":
        cx, cy, cz = 1.0, 1.0, 0.01
        w = e = -cx; s = nn = -cy; bo = t = -cz; dg = 2.0 * (cx + cy + cz)
    elif kind == "convdiff":
        gx, gy, gz = 1.0, 0.5, 0.25
        w, e, s, nn, bo, t = -(1.0 + gx), -1.0, -(1.0 + gy), -1.0, -(1.0 + gz), -1.0
        dg = 6.0 + gx + gy + gz
    elif kind == "varcoef":
        w = e = s = nn = bo = t = dg = 0.0               # per-row coefficients, filled in below
    else:
        raise ValueError(kind)
    N2 = N * N
    rows = np.arange(k_lo * N2, k_hi * N2, dtype=np.int64)
    i = rows % N; j = (rows // N) % N; k = rows // N2
    offs = np.array([-N2, -N, -1, 0, 1, N, N2], dtype=np.int64)
    coef = np.array([bo, s, w, dg, e, nn, t])
    valid = np.stack([k > 0, j > 0, i > 0, np.ones_like(i, bool), i < N - 1, j < N - 1, k < N - 1], axis=1)
    cols = rows[:, None] + offs[None, :]
    vals = np.broadcast_to(coef, cols.shape)
    if kind == "varcoef":
        # the build's variable-coefficient diffusion operator (generator: kryst_amd/csrc/csr_create.hip,
        # varcoef_weight): edge (r, r + {1, N, N^2}[d]) has weight 0.5 + U(splitmix64(0xD1FF, counter 3 r + d)); off-diagonals are
        # -w, the diagonal is the sum in direction order from 0.0 of the six incident weights, 1.0 for a neighbour outside the box
        def wgt(r, d):
            return 0.5 + splitmix64_uniform_at(0xD1FF, 3 * r + d)
        w6 = [np.where(valid[:, 0], wgt(rows - N2, 2), 1.0), np.where(valid[:, 1], wgt(rows - N, 1), 1.0),
              np.where(valid[:, 2], wgt(rows - 1, 0), 1.0), None,
              np.where(valid[:, 4], wgt(rows, 0), 1.0), np.where(valid[:, 5], wgt(rows, 1), 1.0), np.where(valid[:, 6], wgt(rows, 2), 1.0)]
        vals = np.empty(cols.shape)
        dsum = np.zeros(len(rows))
        for q in (0, 1, 2, 4, 5, 6):
            dsum = dsum + w6[q]
            vals[:, q] = -w6[q]
        vals[:, 3] = dsum
    cnt = valid.sum(axis=1)
    rp = np.zeros(len(rows) + 1, dtype=np.int64)
    np.cumsum(cnt, out=rp[1:])
    return Csr(len(rows), N ** 3, rp, cols[valid], vals[valid], check=False)


def tridiag(n, lower, diag, upper):
    """tests/preconditioner_integration.rs:16-57 builders as sparse CSR (explicit zeros not stored)."""
    a = np.zeros((n, n))
    for i in range(n):
        a[i, i] = diag
        if i > 0:
            a[i, i - 1] = lower
        if i + 1 < n:
            a[i, i + 1] = upper
    return a


def splitmix64_uniform_at(seed, counters):
    """uniform [0,1) from splitmix64(seed, counter) for an array of counters (negative counters wrap like uint64: never used)."""
    with np.errstate(over="ignore"):
        z = np.uint64(seed) + (np.asarray(counters).astype(np.uint64) + np.uint64(1)) * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return (z >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def splitmix64_uniform(seed, n):
    """b[i] = uniform [0,1) from splitmix64(seed + counter i) (SURVEY 8d secondary RHS)."""
    with np.errstate(over="ignore"):
        z = (np.uint64(seed) + (np.arange(1, n + 1, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)))
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return (z >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)
