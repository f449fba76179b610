"""ctypes binding of libkryst_hip.so (include/kryst_hip.h).  No torch, no CPU fallback: if the library is
missing or there is no GPU, calls fail loudly."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# KRYST_HIP_LIB: measurement knob, selects a side-by-side build (make VARIANT=...) of the same library
LIB_PATH = os.environ.get("KRYST_HIP_LIB") or os.path.join(_HERE, "lib", "libkryst_hip.so")

OK = 0
ERR_NAMES = {1: "FactorError", 2: "SolveError", 3: "IndefiniteMatrix", 4: "IndefinitePreconditioner",
             5: "ZeroPivot", 6: "Unsupported", 100: "HipError", 101: "RcclError", 102: "ArgumentError",
             103: "CsrError", 104: "ContextBusy"}

c_dp = C.POINTER(C.c_double)
c_i64p = C.POINTER(C.c_int64)
c_u64p = C.POINTER(C.c_uint64)
c_i32p = C.POINTER(C.c_int32)
Handle = C.c_void_p
MONITOR = C.CFUNCTYPE(None, C.c_int64, C.c_double, C.c_void_p)


class Params(C.Structure):
    _fields_ = [("tol", C.c_double), ("max_iters", C.c_int64), ("restart", C.c_int32),
                ("precond_side", C.c_int32), ("norm_type", C.c_int32), ("single_reduction", C.c_int32),
                ("has_radius", C.c_int32), ("radius", C.c_double),
                ("has_obj_target", C.c_int32), ("obj_target", C.c_double), ("check_every", C.c_int32)]


class Stats(C.Structure):
    _fields_ = [("iterations", C.c_int64), ("final_residual", C.c_double), ("converged", C.c_int32)]


_SOLVE_TAIL = [Handle, Handle, C.POINTER(Params), C.POINTER(Stats), c_dp, C.c_int64, c_i64p, MONITOR, C.c_void_p]

# name -> (restype, argtypes): every symbol include/kryst_hip.h declares
SIGNATURES = {
    "kryst_hip_last_error": (C.c_char_p, []),
    "kryst_hip_last_error_row": (C.c_int64, []),
    "kryst_hip_abi_version": (C.c_int32, []),
    "kryst_device_count": (C.c_int32, [C.POINTER(C.c_int32)]),
    "kryst_reduce_spec": (None, [c_i32p, c_i32p, c_i32p]),
    "kryst_ctx_create": (C.c_int32, [C.c_int32, C.POINTER(Handle)]),
    "kryst_comm_unique_id": (C.c_int32, [C.c_void_p]),
    "kryst_ctx_create_dist": (C.c_int32, [C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.POINTER(Handle)]),
    "kryst_ctx_destroy": (C.c_int32, [Handle]),
    "kryst_ctx_synchronize": (C.c_int32, [Handle]),
    "kryst_ctx_rank": (C.c_int32, [Handle, c_i32p, c_i32p]),
    "kryst_comm_barrier": (C.c_int32, [Handle]),
    "kryst_comm_all_reduce": (C.c_int32, [Handle, C.c_double, c_dp]),
    "kryst_ctx_scalar_reduce": (C.c_int32, [Handle, C.c_int32, c_i32p]),
    "kryst_csr_halo_mode": (C.c_int32, [Handle, C.c_int32, c_i32p]),
    "kryst_ctx_trim": (C.c_int32, [Handle, c_i64p]),
    "kryst_phase_timing_begin": (C.c_int32, [Handle]),
    "kryst_phase_timing_end": (C.c_int32, [Handle, c_dp, C.c_int32]),
    "kryst_phase_count": (C.c_int32, []),
    "kryst_phase_name": (C.c_char_p, [C.c_int32]),
    "kryst_ctx_timer_start": (C.c_int32, [Handle]),
    "kryst_ctx_timer_stop": (C.c_int32, [Handle, c_dp]),
    "kryst_vec_create": (C.c_int32, [Handle, C.c_int64, C.POINTER(Handle)]),
    "kryst_vec_destroy": (C.c_int32, [Handle]),
    "kryst_vec_len": (C.c_int32, [Handle, c_i64p]),
    "kryst_vec_upload": (C.c_int32, [Handle, c_dp, C.c_int64]),
    "kryst_vec_download": (C.c_int32, [Handle, c_dp, C.c_int64]),
    "kryst_vec_fill": (C.c_int32, [Handle, C.c_double]),
    "kryst_vec_copy": (C.c_int32, [Handle, Handle]),
    "kryst_vec_fill_splitmix": (C.c_int32, [Handle, C.c_uint64, C.c_int64]),
    "kryst_csr_create": (C.c_int32, [Handle, C.c_int64, C.c_int64, c_u64p, c_u64p, c_dp, C.POINTER(Handle)]),
    "kryst_csr_create_i32": (C.c_int32, [Handle, C.c_int64, C.c_int64, c_i64p, c_i32p, c_dp, C.POINTER(Handle)]),
    "kryst_csr_create_dist": (C.c_int32, [Handle, C.c_int64, c_i64p, c_i64p, c_i64p, c_dp, C.POINTER(Handle)]),
    "kryst_csr_create_stencil7": (C.c_int32, [Handle, C.c_int32, C.c_int32, C.POINTER(Handle)]),
    "kryst_csr_destroy": (C.c_int32, [Handle]),
    "kryst_csr_shape": (C.c_int32, [Handle, c_i64p, c_i64p, c_i64p]),
    "kryst_csr_download": (C.c_int32, [Handle, c_i64p, c_i32p, c_dp]),
    "kryst_csr_placement_info": (C.c_int32, [Handle, c_i32p, c_i32p, c_dp]),
    "kryst_spmv": (C.c_int32, [Handle, Handle, Handle]),
    "kryst_spmv_host": (C.c_int32, [Handle, c_dp, C.c_int64, c_dp, C.c_int64]),
    "kryst_csr_encoding": (C.c_int32, [Handle, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "kryst_csr_tile_order": (C.c_int32, [Handle, C.POINTER(C.c_int64)]),
    "kryst_csr_pattern_info": (C.c_int32, [Handle, C.POINTER(C.c_int64)]),
    "kryst_bench_spmv": (C.c_int32, [Handle, Handle, Handle, C.c_int32, C.c_int32, c_dp]),
    "kryst_bench_streams": (C.c_int32, [Handle, C.c_int64, C.c_int64, C.c_int32, C.c_int32, c_dp]),
    "kryst_bench_csr_skeleton": (C.c_int32, [Handle, Handle, Handle, C.c_int32, c_dp]),
    "kryst_bench_spmv_fused": (C.c_int32, [Handle, Handle, Handle, C.c_int32, c_dp]),
    "kryst_bench_poison_lds": (C.c_int32, [Handle]),
    "kryst_dot": (C.c_int32, [Handle, Handle, c_dp]),
    "kryst_norm": (C.c_int32, [Handle, c_dp]),
    "kryst_axpy": (C.c_int32, [C.c_double, Handle, Handle]),
    "kryst_aypx": (C.c_int32, [C.c_double, Handle, Handle]),
    "kryst_sub": (C.c_int32, [Handle, Handle, Handle]),
    "kryst_pc_identity": (C.c_int32, [Handle, C.POINTER(Handle)]),
    "kryst_pc_jacobi": (C.c_int32, [Handle, C.POINTER(Handle)]),
    "kryst_pc_ilu0": (C.c_int32, [Handle, C.c_int32, C.POINTER(Handle)]),
    "kryst_pc_ilup": (C.c_int32, [Handle, C.c_int32, C.POINTER(Handle)]),
    "kryst_pc_ilut": (C.c_int32, [Handle, C.c_int32, C.c_double, C.POINTER(Handle)]),
    "kryst_pc_chebyshev_stub": (C.c_int32, [Handle, C.c_int32, C.POINTER(Handle)]),
    "kryst_pc_chebyshev": (C.c_int32, [Handle, C.c_double, C.c_double, C.c_int32, C.POINTER(Handle)]),
    "kryst_pc_approx_inverse": (C.c_int32, [Handle, C.POINTER(Handle)]),
    "kryst_pc_apply": (C.c_int32, [Handle, Handle, Handle]),
    "kryst_pc_destroy": (C.c_int32, [Handle]),
    "kryst_bench_pc_apply": (C.c_int32, [Handle, Handle, Handle, C.c_int32, c_dp]),
    "kryst_pc_ilu_info": (C.c_int32, [Handle, c_i64p, C.c_int32]),
    "kryst_apply_chebyshev": (C.c_int32, [Handle, Handle, Handle, C.c_double, C.c_double, C.c_int64]),
    "kryst_cg_solve": (C.c_int32, [c_dp, c_dp, C.c_int64] + _SOLVE_TAIL),
    "kryst_pcg_solve": (C.c_int32, [c_dp, c_dp, C.c_int64] + _SOLVE_TAIL),
    "kryst_gmres_solve": (C.c_int32, [c_dp, c_dp, C.c_int64] + _SOLVE_TAIL),
    "kryst_bicgstab_solve": (C.c_int32, [c_dp, c_dp, C.c_int64] + _SOLVE_TAIL),
    "kryst_cg_solve_dev": (C.c_int32, [Handle, Handle] + _SOLVE_TAIL),
    "kryst_pcg_solve_dev": (C.c_int32, [Handle, Handle] + _SOLVE_TAIL),
    "kryst_gmres_solve_dev": (C.c_int32, [Handle, Handle] + _SOLVE_TAIL),
    "kryst_cgs_solve": (C.c_int32, [c_dp, c_dp, C.c_int64] + _SOLVE_TAIL),
    "kryst_tfqmr_solve": (C.c_int32, [c_dp, c_dp, C.c_int64] + _SOLVE_TAIL),
    "kryst_cgs_solve_dev": (C.c_int32, [Handle, Handle] + _SOLVE_TAIL),
    "kryst_tfqmr_solve_dev": (C.c_int32, [Handle, Handle] + _SOLVE_TAIL),
    "kryst_fgmres_solve": (C.c_int32, [c_dp, c_dp, C.c_int64, C.c_int32, C.c_double, C.c_int32] + _SOLVE_TAIL),
    "kryst_fgmres_solve_dev": (C.c_int32, [Handle, Handle, C.c_int32, C.c_double, C.c_int32] + _SOLVE_TAIL),
    "kryst_bicgstab_solve_dev": (C.c_int32, [Handle, Handle] + _SOLVE_TAIL),
    "kryst_bicgstab_rpc_solve_dev": (C.c_int32, [Handle, Handle] + _SOLVE_TAIL),
    "kryst_session_begin": (C.c_int32, [C.c_int32, Handle, Handle, Handle, Handle, C.POINTER(Params), C.POINTER(Handle)]),
    "kryst_session_step": (C.c_int32, [Handle, C.c_int64]),
    "kryst_session_end": (C.c_int32, [Handle, C.POINTER(Stats), c_dp, C.c_int64, c_i64p]),
    "kryst_host_stencil7": (C.c_int64, [C.c_int32, C.c_int32, C.c_int32, C.c_int32, c_i64p, c_i64p, c_dp]),
    "kryst_host_partition_rows": (C.c_int32, [C.c_int64, C.c_int32, C.c_int64, c_i64p]),
    "kryst_host_read_petsc_binary": (C.c_int64, [C.c_char_p, c_i64p, c_i64p, c_i64p, c_i64p, c_dp]),
    "kryst_host_read_matrix_market": (C.c_int64, [C.c_char_p, c_i64p, c_i64p, c_i64p, c_i64p, c_dp]),
    "kryst_host_halo_recv_plan": (C.c_int64, [C.c_int32, C.c_int32, c_i64p, c_i64p, c_i64p, c_i64p, c_i64p]),
    "kryst_host_ilup": (C.c_int32, [C.c_int64, c_i64p, c_i32p, c_dp, C.c_int32, C.c_int32, C.c_int64, C.POINTER(Handle)]),
    "kryst_host_ilut": (C.c_int32, [C.c_int64, c_i64p, c_i32p, c_dp, C.c_int32, C.c_double, C.c_int32, C.POINTER(Handle)]),
    "kryst_host_factors_sizes": (C.c_int32, [Handle, c_i64p, c_i64p, c_i64p]),
    "kryst_host_factors_get": (C.c_int32, [Handle, c_i64p, c_i32p, c_dp, c_i64p, c_i32p, c_dp, c_dp]),
    "kryst_host_factors_destroy": (C.c_int32, [Handle]),
    "kryst_host_levels": (C.c_int32, [C.c_int64, c_i64p, c_i32p, C.c_int32, c_i32p, c_i32p]),
}

_lib = None


def lib():
    """Load libkryst_hip.so (built by __graft_entry__.build() / make -C kryst_amd/csrc)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} is missing: the HIP extension is not built (run `make -C kryst_amd/csrc` "
                               "or __graft_entry__.build()).  kryst_amd has no CPU fallback.")
        L = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)          # AttributeError = missing export
            fn.restype, fn.argtypes = res, args
        _lib = L
    return _lib


def source_sha16(only=None):
    """sha256 (first 16 hex digits) over the library's sources (kryst_amd/csrc/*.{hip,h,cpp}, Makefile, include/kryst_hip.h), or
    over the named files of kryst_amd/csrc only: measurements kept under profiles/ carry it, so that bench.py only quotes one
    beside numbers from the same sources."""
    import glob
    import hashlib
    h = hashlib.sha256()
    if only:
        files = [os.path.join(_HERE, "csrc", f) for f in sorted(only)]
    else:
        files = sorted(glob.glob(os.path.join(_HERE, "csrc", "*.hip")) + glob.glob(os.path.join(_HERE, "csrc", "*.h")) +
                       glob.glob(os.path.join(_HERE, "csrc", "*.cpp")) + [os.path.join(_HERE, "csrc", "Makefile"),
                                                                           os.path.join(os.path.dirname(_HERE), "include", "kryst_hip.h")])
    for f in files:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


# what the SpMV kernels and their launch are built from: the PMC traffic of profiles/spmv_traffic.json is stamped with this, so
# that work on other kernels (the triangular solves, the solvers) does not orphan it
SPMV_SOURCES = ("spmv.hip", "csr_create.hip", "csr.h", "common.h", "ew.h", "Makefile")


def spmv_source_sha16():
    return source_sha16(SPMV_SOURCES)


class KError(Exception):
    """Mirror of kryst::error::KError (src/error.rs:6-19) plus the runtime error classes of the C ABI."""

    def __init__(self, code, message="", stats=None, row=None):
        self.code, self.kind, self.stats = code, ERR_NAMES.get(code, f"code {code}"), stats
        self.row = row                 # KError::ZeroPivot(row), src/error.rs:15-16
        super().__init__(f"{self.kind}: {message}" if message else self.kind)


def check(rc, stats=None):
    if rc != OK:
        msg = lib().kryst_hip_last_error()
        raise KError(rc, msg.decode() if msg else "", stats, lib().kryst_hip_last_error_row() if rc == 5 else None)


def device_count():
    """HIP devices this process sees (0 without a GPU)."""
    c = C.c_int32(0)
    check(lib().kryst_device_count(C.byref(c)))
    return c.value
