//! Raw `extern "C"` declarations of include/kryst_hip.h (ABI version 5), one to one.  Everything returns an `i32` status:
//! 0 OK, 1..6 = `KError` (src/error.rs:6-19), >= 100 runtime / argument errors (`kryst_hip_last_error()` has the text).
#![allow(non_camel_case_types, dead_code)]
use std::os::raw::{c_char, c_void};

pub type Ctx = *mut c_void;
pub type Csr = *mut c_void;
pub type Vecd = *mut c_void;
pub type Pc = *mut c_void;
pub type Session = *mut c_void;
pub type HostFactors = *mut c_void;

pub const KRYST_OK: i32 = 0;
pub const KRYST_FACTOR_ERROR: i32 = 1;
pub const KRYST_SOLVE_ERROR: i32 = 2;
pub const KRYST_INDEFINITE_MATRIX: i32 = 3;
pub const KRYST_INDEFINITE_PRECONDITIONER: i32 = 4;
pub const KRYST_ZERO_PIVOT: i32 = 5;
pub const KRYST_UNSUPPORTED: i32 = 6;
pub const KRYST_ERR_HIP: i32 = 100;
pub const KRYST_ERR_RCCL: i32 = 101;
pub const KRYST_ERR_ARG: i32 = 102;
pub const KRYST_ERR_CSR: i32 = 103;
pub const KRYST_ERR_BUSY: i32 = 104;

pub const KRYST_ILU_KRYST_COMPAT: i32 = 0;
pub const KRYST_ILU_ILUP0: i32 = 1;
pub const KRYST_ILU_TRUE_ILU0: i32 = 2;

/// kryst_params_t
#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct Params {
    pub tol: f64,
    pub max_iters: i64,
    pub restart: i32,
    pub precond_side: i32,
    pub norm_type: i32,
    pub single_reduction: i32,
    pub has_radius: i32,
    pub radius: f64,
    pub has_obj_target: i32,
    pub obj_target: f64,
    pub check_every: i32,
}

/// kryst_stats_t
#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct Stats {
    pub iterations: i64,
    pub final_residual: f64,
    pub converged: i32,
}

pub type MonitorFn = Option<unsafe extern "C" fn(iteration: i64, residual: f64, user: *mut c_void)>;

extern "C" {
    pub fn kryst_hip_last_error() -> *const c_char;
    pub fn kryst_hip_last_error_row() -> i64;
    pub fn kryst_hip_abi_version() -> i32;
    pub fn kryst_device_count(count: *mut i32) -> i32;
    pub fn kryst_reduce_spec(t: *mut i32, v: *mut i32, f: *mut i32);

    pub fn kryst_ctx_create(device_id: i32, out: *mut Ctx) -> i32;
    pub fn kryst_comm_unique_id(out128: *mut c_void) -> i32;
    pub fn kryst_ctx_create_dist(device_id: i32, rank: i32, nranks: i32, unique_id128: *const c_void, out: *mut Ctx) -> i32;
    pub fn kryst_ctx_destroy(ctx: Ctx) -> i32;
    pub fn kryst_ctx_synchronize(ctx: Ctx) -> i32;
    pub fn kryst_ctx_rank(ctx: Ctx, rank: *mut i32, nranks: *mut i32) -> i32;
    pub fn kryst_comm_barrier(ctx: Ctx) -> i32;
    pub fn kryst_comm_all_reduce(ctx: Ctx, x: f64, out: *mut f64) -> i32;
    pub fn kryst_ctx_scalar_reduce(ctx: Ctx, mode: i32, active: *mut i32) -> i32;
    pub fn kryst_csr_halo_mode(a: Csr, mode: i32, active: *mut i32) -> i32;
    pub fn kryst_ctx_trim(ctx: Ctx, bytes_released: *mut i64) -> i32;
    pub fn kryst_phase_timing_begin(ctx: Ctx) -> i32;
    pub fn kryst_phase_timing_end(ctx: Ctx, ms: *mut f64, count: i32) -> i32;
    pub fn kryst_phase_count() -> i32;
    pub fn kryst_phase_name(phase: i32) -> *const c_char;
    pub fn kryst_ctx_timer_start(ctx: Ctx) -> i32;
    pub fn kryst_ctx_timer_stop(ctx: Ctx, ms: *mut f64) -> i32;

    pub fn kryst_vec_create(ctx: Ctx, n: i64, out: *mut Vecd) -> i32;
    pub fn kryst_vec_destroy(v: Vecd) -> i32;
    pub fn kryst_vec_len(v: Vecd, n: *mut i64) -> i32;
    pub fn kryst_vec_upload(v: Vecd, host: *const f64, n: i64) -> i32;
    pub fn kryst_vec_download(v: Vecd, host: *mut f64, n: i64) -> i32;
    pub fn kryst_vec_fill(v: Vecd, value: f64) -> i32;
    pub fn kryst_vec_copy(dst: Vecd, src: Vecd) -> i32;
    pub fn kryst_vec_fill_splitmix(v: Vecd, seed: u64, global_offset: i64) -> i32;

    pub fn kryst_csr_create(ctx: Ctx, nrows: i64, ncols: i64, row_ptr: *const u64, col_idx: *const u64, vals: *const f64, out: *mut Csr) -> i32;
    pub fn kryst_csr_create_i32(ctx: Ctx, nrows: i64, ncols: i64, row_ptr: *const i64, col_idx: *const i32, vals: *const f64, out: *mut Csr) -> i32;
    pub fn kryst_csr_create_dist(ctx: Ctx, n_global: i64, row_offsets: *const i64, row_ptr: *const i64, col_idx_global: *const i64,
                                 vals: *const f64, out: *mut Csr) -> i32;
    pub fn kryst_csr_create_stencil7(ctx: Ctx, n: i32, kind: i32, out: *mut Csr) -> i32;
    pub fn kryst_csr_destroy(a: Csr) -> i32;
    pub fn kryst_csr_shape(a: Csr, nrows_local: *mut i64, ncols_global: *mut i64, nnz_local: *mut i64) -> i32;
    pub fn kryst_csr_encoding(a: Csr, encoding: *mut i32, patterns: *mut i32, table_entries: *mut i32) -> i32;
    pub fn kryst_csr_tile_order(a: Csr, info: *mut i64) -> i32;
    pub fn kryst_csr_pattern_info(a: Csr, info: *mut i64) -> i32;
    pub fn kryst_csr_download(a: Csr, row_ptr: *mut i64, col_idx_local: *mut i32, vals: *mut f64) -> i32;
    pub fn kryst_csr_placement_info(a: Csr, tries: *mut i32, chosen: *mut i32, skeleton_ms8: *mut f64) -> i32;

    pub fn kryst_spmv(a: Csr, x: Vecd, y: Vecd) -> i32;
    pub fn kryst_spmv_host(a: Csr, x: *const f64, nx: i64, y: *mut f64, ny: i64) -> i32;
    pub fn kryst_bench_spmv(a: Csr, x: Vecd, y: Vecd, fused_dots: i32, reps: i32, avg_ms: *mut f64) -> i32;
    pub fn kryst_bench_streams(ctx: Ctx, n: i64, stride_bytes: i64, kind: i32, reps: i32, avg_ms: *mut f64) -> i32;
    pub fn kryst_bench_csr_skeleton(a: Csr, x: Vecd, y: Vecd, reps: i32, avg_ms: *mut f64) -> i32;
    pub fn kryst_bench_spmv_fused(a: Csr, x: Vecd, y: Vecd, reps: i32, avg_ms: *mut f64) -> i32;
    pub fn kryst_bench_poison_lds(ctx: Ctx) -> i32;

    pub fn kryst_dot(x: Vecd, y: Vecd, out: *mut f64) -> i32;
    pub fn kryst_norm(x: Vecd, out: *mut f64) -> i32;
    pub fn kryst_axpy(alpha: f64, x: Vecd, y: Vecd) -> i32;
    pub fn kryst_aypx(beta: f64, x: Vecd, y: Vecd) -> i32;
    pub fn kryst_sub(a: Vecd, b: Vecd, out: Vecd) -> i32;

    pub fn kryst_pc_identity(ctx: Ctx, out: *mut Pc) -> i32;
    pub fn kryst_pc_jacobi(a: Csr, out: *mut Pc) -> i32;
    pub fn kryst_pc_ilu0(a: Csr, mode: i32, out: *mut Pc) -> i32;
    pub fn kryst_pc_ilup(a: Csr, fill: i32, out: *mut Pc) -> i32;
    pub fn kryst_pc_ilut(a: Csr, fill: i32, droptol: f64, out: *mut Pc) -> i32;
    pub fn kryst_pc_chebyshev_stub(ctx: Ctx, degree: i32, out: *mut Pc) -> i32;
    pub fn kryst_pc_chebyshev(a: Csr, alpha: f64, beta: f64, degree: i32, out: *mut Pc) -> i32;
    pub fn kryst_pc_approx_inverse(m: Csr, out: *mut Pc) -> i32;
    pub fn kryst_pc_apply(pc: Pc, r: Vecd, z: Vecd) -> i32;
    pub fn kryst_pc_destroy(pc: Pc) -> i32;
    pub fn kryst_bench_pc_apply(pc: Pc, r: Vecd, z: Vecd, reps: i32, avg_ms: *mut f64) -> i32;
    pub fn kryst_pc_ilu_info(pc: Pc, info: *mut i64, count: i32) -> i32;
    pub fn kryst_apply_chebyshev(a: Csr, r: Vecd, z: Vecd, alpha: f64, beta: f64, m: i64) -> i32;
}

/// The tail every solve entry point shares (KRYST_SOLVE_ARGS).
macro_rules! solve_fn {
    ($($name:ident),* ; host) => { extern "C" { $( pub fn $name(b: *const f64, x: *mut f64, n: i64, a: Csr, pc: Pc, params: *const Params,
        stats: *mut Stats, hist: *mut f64, hist_cap: i64, hist_len: *mut i64, monitor: MonitorFn, user: *mut c_void) -> i32; )* } };
    ($($name:ident),* ; dev) => { extern "C" { $( pub fn $name(b: Vecd, x: Vecd, a: Csr, pc: Pc, params: *const Params,
        stats: *mut Stats, hist: *mut f64, hist_cap: i64, hist_len: *mut i64, monitor: MonitorFn, user: *mut c_void) -> i32; )* } };
}
solve_fn!(kryst_cg_solve, kryst_pcg_solve, kryst_gmres_solve, kryst_bicgstab_solve, kryst_cgs_solve, kryst_tfqmr_solve; host);
solve_fn!(kryst_cg_solve_dev, kryst_pcg_solve_dev, kryst_gmres_solve_dev, kryst_bicgstab_solve_dev, kryst_bicgstab_rpc_solve_dev,
          kryst_cgs_solve_dev, kryst_tfqmr_solve_dev; dev);

extern "C" {
    pub fn kryst_fgmres_solve(b: *const f64, x: *mut f64, n: i64, orthog: i32, haptol: f64, preallocate: i32, a: Csr, pc: Pc,
                              params: *const Params, stats: *mut Stats, hist: *mut f64, hist_cap: i64, hist_len: *mut i64,
                              monitor: MonitorFn, user: *mut c_void) -> i32;
    pub fn kryst_fgmres_solve_dev(b: Vecd, x: Vecd, orthog: i32, haptol: f64, preallocate: i32, a: Csr, pc: Pc,
                                  params: *const Params, stats: *mut Stats, hist: *mut f64, hist_cap: i64, hist_len: *mut i64,
                                  monitor: MonitorFn, user: *mut c_void) -> i32;

    pub fn kryst_session_begin(method: i32, b: Vecd, x: Vecd, a: Csr, pc: Pc, params: *const Params, out: *mut Session) -> i32;
    pub fn kryst_session_step(s: Session, k: i64) -> i32;
    pub fn kryst_session_end(s: Session, stats: *mut Stats, hist: *mut f64, hist_cap: i64, hist_len: *mut i64) -> i32;

    pub fn kryst_host_stencil7(n: i32, kind: i32, k_lo: i32, k_hi: i32, row_ptr: *mut i64, col_idx: *mut i64, vals: *mut f64) -> i64;
    pub fn kryst_host_partition_rows(n: i64, nranks: i32, align: i64, row_offsets: *mut i64) -> i32;
    pub fn kryst_host_halo_recv_plan(rank: i32, nranks: i32, row_offsets: *const i64, row_ptr: *const i64, col_idx_global: *const i64,
                                     recv_counts: *mut i64, recv_cols: *mut i64) -> i64;
    pub fn kryst_host_ilup(n: i64, row_ptr: *const i64, col: *const i32, val: *const f64, fill: i32, threads: i32, block: i64, out: *mut HostFactors) -> i32;
    pub fn kryst_host_ilut(n: i64, row_ptr: *const i64, col: *const i32, val: *const f64, fill: i32, droptol: f64, threads: i32, out: *mut HostFactors) -> i32;
    pub fn kryst_host_factors_sizes(f: HostFactors, n: *mut i64, nnz_l: *mut i64, nnz_u: *mut i64) -> i32;
    pub fn kryst_host_factors_get(f: HostFactors, l_ptr: *mut i64, l_col: *mut i32, l_val: *mut f64, u_ptr: *mut i64, u_col: *mut i32, u_val: *mut f64, diag: *mut f64) -> i32;
    pub fn kryst_host_factors_destroy(f: HostFactors) -> i32;
    pub fn kryst_host_levels(n: i64, ptr: *const i64, col: *const i32, forward: i32, level: *mut i32, nlevels: *mut i32) -> i32;
    pub fn kryst_host_read_matrix_market(path: *const c_char, nrows: *mut i64, ncols: *mut i64, row_ptr: *mut i64, col_idx: *mut i64,
                                         vals: *mut f64) -> i64;
    pub fn kryst_host_read_petsc_binary(path: *const c_char, nrows: *mut i64, ncols: *mut i64, row_ptr: *mut i64, col_idx: *mut i64,
                                        vals: *mut f64) -> i64;
}
