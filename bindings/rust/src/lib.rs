//! kryst-hip -- kryst's Krylov inner loop on one MI355X (gfx950), behind kryst's own traits.
//!
//! UNVERIFIED SOURCE: written against kryst v0.5.3 and include/kryst_hip.h (ABI 2) without a Rust toolchain at hand; it has
//! never been compiled.  tests/test_rust_binding_cpu.py checks the parts a text check can (every symbol of the header is
//! declared in `ffi.rs` with the header's argument count, `#[repr(C)]` field order).
//!
//! What is what (reference file:line -> type here):
//!
//! | reference | here |
//! |---|---|
//! | `CsrMatrix::from_csr` src/matrix/sparse.rs:28-46, `SparseMatrix::spmv` :56-67 | [`HipCsrMatrix::from_csr`], `impl MatVec<Vec<f64>>` (operator-level: PCIe both ways) |
//! | `Preconditioner::{setup,apply}` src/preconditioner/mod.rs:8-13 | [`HipJacobi`], [`HipIlu0`], [`HipChebyshev`] (`impl Preconditioner<HipCsrMatrix, Vec<f64>>`) |
//! | `LinearSolver::solve` src/solver/mod.rs:30-52 | [`HipCgSolver`], [`HipPcgSolver`], [`HipGmresSolver`], [`HipBiCgStabSolver`]: the reference structs' public fields and builders, device-resident iteration |
//! | `KspContext::solve_context` src/context/ksp_context.rs:88-148 | [`HipKspContext`] |
//! | `KError` src/error.rs:6-19 | mapped by [`kerr`], `ZeroPivot(row)` from `kryst_hip_last_error_row()` |
//!
//! The solvers receive `pc: Option<&dyn Preconditioner<HipCsrMatrix, Vec<f64>>>` type-erased, like the reference's.  A device
//! preconditioner is recognised by an in-band probe (see [`probe_device_pc`]): no change to kryst's trait is needed.  A
//! preconditioner that is not one of this crate's is refused with `KError::Unsupported` (the C ABI has no host-callback
//! preconditioner; wrapping the reference's own solver around `HipCsrMatrix: MatVec` stays possible and is PCIe-bound).
pub mod ffi;

use std::cell::RefCell;
use std::ffi::CStr;
use std::os::raw::c_void;
use std::rc::Rc;

use kryst::core::traits::{Indexing, MatShape, MatVec};
use kryst::error::KError;
use kryst::preconditioner::Preconditioner;
use kryst::solver::gmres::Preconditioning;
use kryst::solver::LinearSolver;
use kryst::utils::convergence::{Convergence, SolveStats};

/// `KError` of a status code (src/error.rs:6-19).  Codes >= 100 (HIP / RCCL / argument / busy) have no counterpart in the
/// reference, where they would be panics or cannot happen: they surface as `SolveError` with the library's message.
pub fn kerr(code: i32) -> KError {
    let msg = unsafe { CStr::from_ptr(ffi::kryst_hip_last_error()) }.to_string_lossy().into_owned();
    match code {
        ffi::KRYST_FACTOR_ERROR => KError::FactorError(msg),
        ffi::KRYST_SOLVE_ERROR => KError::SolveError(msg),
        ffi::KRYST_INDEFINITE_MATRIX => KError::IndefiniteMatrix,
        ffi::KRYST_INDEFINITE_PRECONDITIONER => KError::IndefinitePreconditioner,
        ffi::KRYST_ZERO_PIVOT => KError::ZeroPivot(unsafe { ffi::kryst_hip_last_error_row() }.max(0) as usize),
        ffi::KRYST_UNSUPPORTED => KError::Unsupported("kryst-hip: operation outside the accelerated path"),
        _ => KError::SolveError(format!("kryst-hip status {code}: {msg}")),
    }
}

fn check(code: i32) -> Result<(), KError> {
    if code == ffi::KRYST_OK { Ok(()) } else { Err(kerr(code)) }
}

// ------------------------------------------------------------------------------------------------------------------ context
struct CtxInner(ffi::Ctx);
impl Drop for CtxInner {
    fn drop(&mut self) {
        unsafe { ffi::kryst_ctx_destroy(self.0) };
    }
}

/// One GPU (one rank).  Replaces the `Comm` objects of src/parallel (`RayonComm` / `MpiComm`, parallel/mod.rs:4-35).
/// One host thread per context; ONE solve at a time per context (a second one returns `KRYST_ERR_BUSY`).
#[derive(Clone)]
pub struct HipContext(Rc<CtxInner>);

impl HipContext {
    pub fn new(device: i32) -> Result<Self, KError> {
        let mut h: ffi::Ctx = std::ptr::null_mut();
        check(unsafe { ffi::kryst_ctx_create(device, &mut h) })?;
        Ok(Self(Rc::new(CtxInner(h))))
    }
    /// `MpiComm::new` (src/parallel/mpi_comm.rs:49-55): rank 0 obtains `unique_id()` and ships the 128 bytes to the other
    /// ranks by any side channel (MPI, a file, a socket).
    pub fn new_dist(device: i32, rank: i32, nranks: i32, unique_id: &[u8; 128]) -> Result<Self, KError> {
        let mut h: ffi::Ctx = std::ptr::null_mut();
        check(unsafe { ffi::kryst_ctx_create_dist(device, rank, nranks, unique_id.as_ptr() as *const c_void, &mut h) })?;
        Ok(Self(Rc::new(CtxInner(h))))
    }
    pub fn unique_id() -> Result<[u8; 128], KError> {
        let mut id = [0u8; 128];
        check(unsafe { ffi::kryst_comm_unique_id(id.as_mut_ptr() as *mut c_void) })?;
        Ok(id)
    }
    /// `Comm::all_reduce` (mpi_comm.rs:116-121): rank results folded in rank order, the same bits on every rank.
    pub fn all_reduce(&self, x: f64) -> Result<f64, KError> {
        let mut out = 0.0;
        check(unsafe { ffi::kryst_comm_all_reduce(self.raw(), x, &mut out) })?;
        Ok(out)
    }
    pub fn barrier(&self) -> Result<(), KError> {
        check(unsafe { ffi::kryst_comm_barrier(self.raw()) })
    }
    /// How the solvers' `DistributedInnerProduct` (core/wrappers.rs:134-156) crosses the ranks: `false` = RCCL all-gather +
    /// rank-ordered fold (default), `true` = hipIpc mailboxes written and polled by the fold kernel (one launch, no collective,
    /// the same bits).  Collective over the context's ranks.  Returns whether the mailbox path is in use afterwards (it is not
    /// when some rank cannot export or map a mailbox: every rank then stays on RCCL).
    pub fn scalar_reduce_ipc(&self, on: bool) -> Result<bool, KError> {
        let mut active = 0i32;
        let rc = unsafe { ffi::kryst_ctx_scalar_reduce(self.raw(), if on { 1 } else { 0 }, &mut active) };
        if rc != 0 && rc != 6 { check(rc)?; }
        Ok(active != 0)
    }
    pub fn synchronize(&self) -> Result<(), KError> {
        check(unsafe { ffi::kryst_ctx_synchronize(self.raw()) })
    }
    fn raw(&self) -> ffi::Ctx {
        (self.0).0
    }
}

// ------------------------------------------------------------------------------------------------------------------ operator
/// `CsrMatrix<f64>` (src/matrix/sparse.rs:22-46) resident in HBM.
pub struct HipCsrMatrix {
    ctx: HipContext,
    h: ffi::Csr,
    nrows: usize,
    ncols: usize,
}

impl HipCsrMatrix {
    /// `CsrMatrix::from_csr(nrows, ncols, row_ptr, col_idx, values)`: the same validation as `new_checked` (sparse.rs:36-42:
    /// monotone row_ptr, in-range strictly ascending columns), reported as `Err` instead of a panic.
    pub fn from_csr(ctx: &HipContext, nrows: usize, ncols: usize, row_ptr: &[usize], col_idx: &[usize], values: &[f64]) -> Result<Self, KError> {
        if row_ptr.len() != nrows + 1 || col_idx.len() != values.len() || row_ptr.last().copied().unwrap_or(0) != values.len() {
            return Err(KError::SolveError("from_csr: array lengths do not describe a CSR matrix".into()));
        }
        const _: () = assert!(std::mem::size_of::<usize>() == 8);       // the ABI takes the reference's usize arrays as uint64
        let mut h: ffi::Csr = std::ptr::null_mut();
        check(unsafe {
            ffi::kryst_csr_create(ctx.raw(), nrows as i64, ncols as i64, row_ptr.as_ptr() as *const u64, col_idx.as_ptr() as *const u64,
                                  values.as_ptr(), &mut h)
        })?;
        Ok(Self { ctx: ctx.clone(), h, nrows, ncols })
    }
    pub fn context(&self) -> &HipContext {
        &self.ctx
    }
    /// The halo exchange of a row-partitioned operator (the neighbour exchange `src/parallel/mpi_comm.rs:133-143` leaves as a TODO):
    /// `false` = grouped ncclSend / ncclRecv (default), `true` = direct peer stores into hipIpc-mapped landing buffers (no collective
    /// launch, the same bits).  Collective over the context's ranks.  Returns whether the peer-store path is in use afterwards.
    pub fn halo_peer_stores(&self, on: bool) -> Result<bool, KError> {
        let mut active = 0i32;
        let rc = unsafe { ffi::kryst_csr_halo_mode(self.h, if on { 1 } else { 0 }, &mut active) };
        if rc != 0 && rc != 6 { check(rc)?; }
        Ok(active != 0)
    }
    /// `SparseMatrix::spmv` (sparse.rs:56-67) on host slices.
    pub fn spmv(&self, x: &[f64], y: &mut [f64]) {
        assert_eq!(x.len(), self.ncols);                                  // sparse.rs:57
        assert_eq!(y.len(), self.nrows);                                  // sparse.rs:58
        let rc = unsafe { ffi::kryst_spmv_host(self.h, x.as_ptr(), x.len() as i64, y.as_mut_ptr(), y.len() as i64) };
        assert_eq!(rc, 0, "kryst-hip spmv: {:?}", kerr(rc));
    }
}

impl Drop for HipCsrMatrix {
    fn drop(&mut self) {
        unsafe { ffi::kryst_csr_destroy(self.h) };
    }
}

/// Operator-level drop-in (src/core/traits.rs:4-7).  Every call moves x up and y down the PCIe link: correctness plumbing
/// (e.g. the reference's `Jacobi::setup`, jacobi.rs:53-67, or its own solvers) -- the fast path is the solver-level impls.
impl MatVec<Vec<f64>> for HipCsrMatrix {
    fn matvec(&self, x: &Vec<f64>, y: &mut Vec<f64>) {
        self.spmv(x, y)
    }
}
impl Indexing for HipCsrMatrix {
    fn nrows(&self) -> usize {
        self.nrows
    }
}
impl MatShape for HipCsrMatrix {
    fn nrows(&self) -> usize {
        self.nrows
    }
    fn ncols(&self) -> usize {
        self.ncols
    }
}

// ------------------------------------------------------------------------------------------------------------------ preconditioners
/// How a solver finds the device handle behind a type-erased `&dyn Preconditioner<HipCsrMatrix, Vec<f64>>`: it calls
/// `apply(&[], &mut [PROBE, 0.0])`.  This crate's preconditioners recognise the empty `r` with the magic word in `z[0]` and
/// answer with their handle's bits in `z[1]`; anybody else's `apply` sees an empty residual, loops zero times (or returns an
/// error) and leaves `z[1]` alone.
const PROBE: u64 = 0x4B52_5953_5448_4950; // "KRYSTHIP"

pub fn probe_device_pc(pc: &dyn Preconditioner<HipCsrMatrix, Vec<f64>>) -> Option<ffi::Pc> {
    let r: Vec<f64> = Vec::new();
    let mut z = vec![f64::from_bits(PROBE), 0.0];
    let _ = pc.apply(&r, &mut z);
    let bits = z[1].to_bits();
    if z[0].to_bits() == PROBE && bits != 0 { Some(bits as usize as ffi::Pc) } else { None }
}

struct PcHandle(ffi::Pc);
impl Drop for PcHandle {
    fn drop(&mut self) {
        if !self.0.is_null() {
            unsafe { ffi::kryst_pc_destroy(self.0) };
        }
    }
}

fn pc_apply_host(ctx: &HipContext, pc: ffi::Pc, r: &Vec<f64>, z: &mut Vec<f64>) -> Result<(), KError> {
    if r.is_empty() && z.len() == 2 && z[0].to_bits() == PROBE {
        z[1] = f64::from_bits(pc as usize as u64);                        // the solver's probe (probe_device_pc)
        return Ok(());
    }
    if pc.is_null() {
        return Err(KError::SolveError("preconditioner used before setup".into()));
    }
    assert_eq!(r.len(), z.len());
    let n = r.len() as i64;
    let (mut dr, mut dz): (ffi::Vecd, ffi::Vecd) = (std::ptr::null_mut(), std::ptr::null_mut());
    unsafe {
        check(ffi::kryst_vec_create(ctx.raw(), n, &mut dr))?;
        let mut rc = ffi::kryst_vec_create(ctx.raw(), n, &mut dz);
        if rc == 0 { rc = ffi::kryst_vec_upload(dr, r.as_ptr(), n); }
        if rc == 0 { rc = ffi::kryst_pc_apply(pc, dr, dz); }
        if rc == 0 { rc = ffi::kryst_vec_download(dz, z.as_mut_ptr(), n); }
        ffi::kryst_vec_destroy(dr);
        ffi::kryst_vec_destroy(dz);
        check(rc)
    }
}

macro_rules! device_pc {
    ($(#[$doc:meta])* $name:ident { $($field:ident : $ty:ty = $init:expr),* } setup($self_:ident, $a:ident, $out:ident) $body:block) => {
        $(#[$doc])*
        pub struct $name { ctx: Option<HipContext>, h: PcHandle, $(pub $field: $ty),* }
        impl $name {
            fn empty($($field: $ty),*) -> Self { Self { ctx: None, h: PcHandle(std::ptr::null_mut()), $($field),* } }
        }
        impl Preconditioner<HipCsrMatrix, Vec<f64>> for $name {
            fn setup(&mut self, a: &HipCsrMatrix) -> Result<(), KError> {
                let $self_ = &*self;
                let $a = a;
                let mut $out: ffi::Pc = std::ptr::null_mut();
                check(unsafe { $body })?;
                self.h = PcHandle($out);
                self.ctx = Some(a.ctx.clone());
                Ok(())
            }
            fn apply(&self, r: &Vec<f64>, z: &mut Vec<f64>) -> Result<(), KError> {
                match &self.ctx {
                    Some(ctx) => pc_apply_host(ctx, self.h.0, r, z),
                    None => Err(KError::SolveError("preconditioner used before setup".into())),
                }
            }
        }
    };
}

device_pc! {
    /// `Jacobi::new()` + `setup` (src/preconditioner/jacobi.rs:26-95): 1/diag taken from the CSR diagonal (bit-identical to the
    /// reference's n unit-vector matvecs: a_ii * 1 + zeros), inv = d != 0 ? 1/d : 0.
    HipJacobi {} setup(_s, a, out) { ffi::kryst_pc_jacobi(a.h, &mut out) }
}
impl HipJacobi {
    pub fn new() -> Self { Self::empty() }
}

device_pc! {
    /// `Ilu0::new()` exactly as written in src/preconditioner/ilu.rs:59-122 (`mode` 0), `Ilup::new(0)` as written in
    /// ilup.rs:77-167 (`mode` 1), or a textbook ILU(0) on A's pattern (`mode` 2, an extension).
    HipIlu0 { mode: i32 = 0 } setup(s, a, out) { ffi::kryst_pc_ilu0(a.h, s.mode, &mut out) }
}
impl HipIlu0 {
    pub fn new() -> Self { Self::empty(ffi::KRYST_ILU_KRYST_COMPAT) }
    pub fn with_mode(mode: i32) -> Self { Self::empty(mode) }
}

device_pc! {
    /// `Chebyshev::new(degree, lambda_min, lambda_max)` (src/preconditioner/chebyshev.rs:35-70).  With both bounds given the
    /// apply is `apply_chebyshev(a, r, z, lambda_min, lambda_max, degree)` (chebyshev.rs:83-140); without them it is the
    /// reference's stub, which returns `Err(SolveError)` (chebyshev.rs:68-70).
    HipChebyshev { degree: usize = 0, lambda_min: Option<f64> = None, lambda_max: Option<f64> = None } setup(s, a, out) {
        match (s.lambda_min, s.lambda_max) {
            (Some(lo), Some(hi)) => ffi::kryst_pc_chebyshev(a.h, lo, hi, s.degree as i32, &mut out),
            _ => ffi::kryst_pc_chebyshev_stub(a.ctx.raw(), s.degree as i32, &mut out),
        }
    }
}
impl HipChebyshev {
    pub fn new(degree: usize, lambda_min: Option<f64>, lambda_max: Option<f64>) -> Self { Self::empty(degree, lambda_min, lambda_max) }
}

// ------------------------------------------------------------------------------------------------------------------ solvers
/// `CgNormType` (src/solver/cg.rs:35, the same enum again in pcg.rs:25).
#[derive(Clone, Copy, Debug, PartialEq, Eq)]
pub enum CgNormType { Preconditioned, Unpreconditioned, Natural, None }

type Monitor = Option<Box<dyn FnMut(usize, f64)>>;

/// C trampoline of `with_monitor` (cg.rs:84-88): the library fires it on the calling thread, in order, while the device iterates.
unsafe extern "C" fn monitor_trampoline(iteration: i64, residual: f64, user: *mut c_void) {
    let cell = &*(user as *const RefCell<&mut dyn FnMut(usize, f64)>);
    (cell.borrow_mut())(iteration as usize, residual);
}

type HostSolveFn = unsafe extern "C" fn(*const f64, *mut f64, i64, ffi::Csr, ffi::Pc, *const ffi::Params, *mut ffi::Stats, *mut f64, i64,
                                        *mut i64, ffi::MonitorFn, *mut c_void) -> i32;

struct Common<'a> {
    params: ffi::Params,
    monitor: Option<&'a mut dyn FnMut(usize, f64)>,
    history: Option<&'a mut Vec<f64>>,
}

fn device_solve(f: HostSolveFn, a: &HipCsrMatrix, pc: Option<&dyn Preconditioner<HipCsrMatrix, Vec<f64>>>, uses_pc: bool, b: &Vec<f64>,
                x: &mut Vec<f64>, c: Common) -> Result<SolveStats<f64>, KError> {
    assert_eq!(b.len(), x.len());
    let pch = match pc {
        Some(p) if uses_pc => probe_device_pc(p).ok_or(KError::Unsupported(
            "kryst-hip: the preconditioner is not a device preconditioner of this crate (HipJacobi / HipIlu0 / HipChebyshev)"))?,
        _ => std::ptr::null_mut(),                                        // CgSolver / BiCgStabSolver ignore pc (cg.rs:115, bicgstab.rs:70)
    };
    let cap = (c.params.max_iters.max(0) as usize).saturating_add(c.params.restart.max(1) as usize + 8).min((1usize << 22) + 8);
    let mut hist = vec![0.0f64; cap];
    let mut len: i64 = 0;
    let mut st = ffi::Stats::default();
    let rc = match c.monitor {
        Some(m) => {
            let cell: RefCell<&mut dyn FnMut(usize, f64)> = RefCell::new(m);
            unsafe { f(b.as_ptr(), x.as_mut_ptr(), b.len() as i64, a.h, pch, &c.params, &mut st, hist.as_mut_ptr(), cap as i64, &mut len,
                       Some(monitor_trampoline), &cell as *const _ as *mut c_void) }
        }
        None => unsafe { f(b.as_ptr(), x.as_mut_ptr(), b.len() as i64, a.h, pch, &c.params, &mut st, hist.as_mut_ptr(), cap as i64, &mut len,
                           None, std::ptr::null_mut()) },
    };
    if let Some(h) = c.history {
        h.extend_from_slice(&hist[..(len.max(0) as usize).min(cap)]);     // residual_history.push (cg.rs:140,263; pcg.rs:146,199)
    }
    check(rc)?;
    Ok(SolveStats { iterations: st.iterations as usize, final_residual: st.final_residual, converged: st.converged != 0 })
}

fn base_params(conv: &Convergence<f64>) -> ffi::Params {
    ffi::Params { tol: conv.tol, max_iters: conv.max_iters as i64, restart: 0, precond_side: 1, norm_type: 1, single_reduction: 0,
                  has_radius: 0, radius: 0.0, has_obj_target: 0, obj_target: 0.0, check_every: 0 }
}

macro_rules! cg_like {
    ($(#[$doc:meta])* $name:ident, $entry:path, uses_pc = $uses_pc:expr) => {
        $(#[$doc])*
        pub struct $name {
            pub conv: Convergence<f64>,
            pub norm_type: CgNormType,
            pub single_reduction: bool,
            pub radius: Option<f64>,
            pub obj_target: Option<f64>,
            pub monitor: Monitor,
            pub residual_history: Vec<f64>,
            /// how many iterations the host enqueues between two looks at the device (and two rounds of monitor callbacks);
            /// 0 = the library's default (8).  The device stops at the exact reference iteration regardless.
            pub check_every: i32,
        }
        impl $name {
            pub fn new(tol: f64, max_iters: usize) -> Self {
                Self { conv: Convergence { tol, max_iters }, norm_type: CgNormType::Unpreconditioned, single_reduction: false, radius: None,
                       obj_target: None, monitor: None, residual_history: Vec::new(), check_every: 0 }
            }
            pub fn with_norm(mut self, norm_type: CgNormType) -> Self { self.norm_type = norm_type; self }
            pub fn with_single_reduction(mut self, flag: bool) -> Self { self.single_reduction = flag; self }
            pub fn with_radius(mut self, radius: f64) -> Self { self.radius = Some(radius); self }
            pub fn with_obj_target(mut self, obj: f64) -> Self { self.obj_target = Some(obj); self }
            pub fn with_monitor<F>(mut self, f: F) -> Self where F: FnMut(usize, f64) + 'static { self.monitor = Some(Box::new(f)); self }
            pub fn clear_history(&mut self) { self.residual_history.clear(); }
        }
        impl LinearSolver<HipCsrMatrix, Vec<f64>> for $name {
            type Error = KError;
            type Scalar = f64;
            fn solve(&mut self, a: &HipCsrMatrix, pc: Option<&dyn Preconditioner<HipCsrMatrix, Vec<f64>>>, b: &Vec<f64>, x: &mut Vec<f64>)
                -> Result<SolveStats<f64>, KError> {
                let mut p = base_params(&self.conv);
                p.norm_type = match self.norm_type { CgNormType::Preconditioned => 0, CgNormType::Unpreconditioned => 1, CgNormType::Natural => 2, CgNormType::None => 3 };
                p.single_reduction = self.single_reduction as i32;
                if let Some(r) = self.radius { p.has_radius = 1; p.radius = r; }
                if let Some(o) = self.obj_target { p.has_obj_target = 1; p.obj_target = o; }
                p.check_every = self.check_every;
                let mon: Option<&mut dyn FnMut(usize, f64)> = match self.monitor.as_mut() { Some(m) => Some(m.as_mut()), None => None };
                device_solve($entry, a, pc, $uses_pc, b, x, Common { params: p, monitor: mon, history: Some(&mut self.residual_history) })
            }
        }
    };
}

cg_like! {
    /// `CgSolver<f64>` (src/solver/cg.rs:40-93,114-288) on the device: same fields, same builders; `pc` is ignored like the
    /// reference's (cg.rs:115); trust-region (cg.rs:177-202) and objective-target (cg.rs:231-252) exits included.
    HipCgSolver, ffi::kryst_cg_solve, uses_pc = false
}
cg_like! {
    /// `PcgSolver<f64>` (src/solver/pcg.rs:31-91,114-222): res0 = sqrt(|r0.z0|) against ||r||_2 as written (pcg.rs:134 vs :192);
    /// `radius` / `obj_target` are accepted and never read, like the reference's.
    HipPcgSolver, ffi::kryst_pcg_solve, uses_pc = true
}

/// `GmresSolver<f64>` (src/solver/gmres.rs:38-60,216-402): restarted GMRES(m), double modified Gram-Schmidt, Left (default) /
/// Right / None preconditioning, the reference's happy-breakdown and Left-mode quirks included.
pub struct HipGmresSolver {
    pub restart: usize,
    pub conv: Convergence<f64>,
    pub preconditioning: Preconditioning,
    /// LABELLED EXTENSION (not in the reference): with `Preconditioning::Left`, run the textbook left-preconditioned GMRES
    /// (kryst_hip.h: precond_side 3 -- Arnoldi on M^-1 A from M^-1 r0, Gram-Schmidt against V) instead of the reference's Left arm
    /// (gmres.rs:240-247,279-307), which orthogonalises against an un-normalised Z[0] and stagnates on BASELINE config 3.
    pub textbook_left: bool,
}
impl HipGmresSolver {
    pub fn new(restart: usize, tol: f64, max_iters: usize) -> Self {
        Self { restart, conv: Convergence { tol, max_iters }, preconditioning: Preconditioning::Left, textbook_left: false }
    }
    pub fn with_preconditioning(mut self, mode: Preconditioning) -> Self { self.preconditioning = mode; self }
    pub fn with_textbook_left(mut self, flag: bool) -> Self { self.textbook_left = flag; self }
}
impl LinearSolver<HipCsrMatrix, Vec<f64>> for HipGmresSolver {
    type Error = KError;
    type Scalar = f64;
    fn solve(&mut self, a: &HipCsrMatrix, pc: Option<&dyn Preconditioner<HipCsrMatrix, Vec<f64>>>, b: &Vec<f64>, x: &mut Vec<f64>)
        -> Result<SolveStats<f64>, KError> {
        let mut p = base_params(&self.conv);
        p.restart = self.restart as i32;
        p.precond_side = match self.preconditioning { Preconditioning::None => 0, Preconditioning::Left => if self.textbook_left { 3 } else { 1 }, Preconditioning::Right => 2 };
        // the reference unwraps the preconditioner in the Left / Right branches (gmres.rs:245): None there is a panic, here too
        if pc.is_none() && p.precond_side != 0 { panic!("GMRES with Left/Right preconditioning needs a preconditioner (gmres.rs:245)"); }
        device_solve(ffi::kryst_gmres_solve, a, pc, true, b, x, Common { params: p, monitor: None, history: None })
    }
}

/// `BiCgStabSolver<f64>` (src/solver/bicgstab.rs:38-50,69-293): absolute tolerance, breakdowns `break`, `pc` ignored
/// (bicgstab.rs:70).
pub struct HipBiCgStabSolver {
    pub conv: Convergence<f64>,
}
impl HipBiCgStabSolver {
    pub fn new(tol: f64, max_iters: usize) -> Self { Self { conv: Convergence { tol, max_iters } } }
}
impl LinearSolver<HipCsrMatrix, Vec<f64>> for HipBiCgStabSolver {
    type Error = KError;
    type Scalar = f64;
    fn solve(&mut self, a: &HipCsrMatrix, pc: Option<&dyn Preconditioner<HipCsrMatrix, Vec<f64>>>, b: &Vec<f64>, x: &mut Vec<f64>)
        -> Result<SolveStats<f64>, KError> {
        device_solve(ffi::kryst_bicgstab_solve, a, pc, false, b, x, Common { params: base_params(&self.conv), monitor: None, history: None })
    }
}

// ------------------------------------------------------------------------------------------------------------------ KspContext
/// `SolverKind` (src/context/ksp_context.rs:25-50), the kinds inside the accelerated path.
#[derive(Debug, Clone, Copy, PartialEq, Eq)]
pub enum HipSolverKind { Cg, Pcg, GmresLeft, GmresRight, Bicgstab }

/// `KspContext` (src/context/ksp_context.rs:54-148) for a device operator: `solve_context(b, x)` builds the solver of `kind` with
/// (`tol`, `max_it`, `restart`) and runs it with `pc`, exactly as the reference's match does (:101-131).
pub struct HipKspContext {
    pub kind: HipSolverKind,
    pub a: HipCsrMatrix,
    pub pc: Option<Box<dyn Preconditioner<HipCsrMatrix, Vec<f64>>>>,
    pub tol: f64,
    pub max_it: usize,
    pub restart: usize,
}
impl HipKspContext {
    pub fn solve_context(&mut self, b: &Vec<f64>, x: &mut Vec<f64>) -> Result<SolveStats<f64>, KError> {
        let pc = self.pc.as_deref();
        match self.kind {
            HipSolverKind::Cg => HipCgSolver::new(self.tol, self.max_it).solve(&self.a, pc, b, x),
            HipSolverKind::Pcg => HipPcgSolver::new(self.tol, self.max_it).solve(&self.a, pc, b, x),
            HipSolverKind::GmresLeft => HipGmresSolver::new(self.restart, self.tol, self.max_it).with_preconditioning(Preconditioning::Left).solve(&self.a, pc, b, x),
            HipSolverKind::GmresRight => HipGmresSolver::new(self.restart, self.tol, self.max_it).with_preconditioning(Preconditioning::Right).solve(&self.a, pc, b, x),
            HipSolverKind::Bicgstab => HipBiCgStabSolver::new(self.tol, self.max_it).solve(&self.a, pc, b, x),
        }
    }
}

#[cfg(test)]
mod tests {
    //! The reference's own known answers through the device path (needs an MI355X and libkryst_hip.so).
    use super::*;

    fn tridiag(n: usize) -> (Vec<usize>, Vec<usize>, Vec<f64>) {
        let (mut rp, mut ci, mut va) = (vec![0usize], Vec::new(), Vec::new());
        for i in 0..n {
            if i > 0 { ci.push(i - 1); va.push(-1.0); }
            ci.push(i); va.push(2.0);
            if i + 1 < n { ci.push(i + 1); va.push(-1.0); }
            rp.push(ci.len());
        }
        (rp, ci, va)
    }

    #[test]
    fn spmv_known_answer() {                                              // src/matrix/sparse.rs:121-144
        let ctx = HipContext::new(0).unwrap();
        let a = HipCsrMatrix::from_csr(&ctx, 2, 3, &[0, 2, 4], &[0, 1, 1, 2], &[1.0, 2.0, 3.0, 4.0]).unwrap();
        let mut y = vec![0.0; 2];
        a.matvec(&vec![1.0; 3], &mut y);
        assert_eq!(y, vec![3.0, 7.0]);
    }

    #[test]
    fn cg_2x2() {                                                         // src/solver/cg.rs:310-323
        let ctx = HipContext::new(0).unwrap();
        let a = HipCsrMatrix::from_csr(&ctx, 2, 2, &[0, 2, 4], &[0, 1, 0, 1], &[4.0, 1.0, 1.0, 3.0]).unwrap();
        let mut x = vec![0.0; 2];
        let st = HipCgSolver::new(1e-10, 20).solve(&a, None, &vec![1.0, 2.0], &mut x).unwrap();
        assert!(st.converged);
        assert!((x[0] - 0.09090909090909091).abs() < 1e-8 && (x[1] - 0.6363636363636364).abs() < 1e-8);
    }

    #[test]
    fn pcg_jacobi_tridiag() {                                             // tests/preconditioner_integration.rs:126-150
        let ctx = HipContext::new(0).unwrap();
        let n = 10;
        let (rp, ci, va) = tridiag(n);
        let a = HipCsrMatrix::from_csr(&ctx, n, n, &rp, &ci, &va).unwrap();
        let mut b = vec![0.0; n];
        a.matvec(&vec![1.0; n], &mut b);
        let mut pc = HipJacobi::new();
        pc.setup(&a).unwrap();
        let mut x = vec![0.0; n];
        let seen = std::rc::Rc::new(RefCell::new(Vec::new()));
        let seen2 = seen.clone();
        let mut s = HipPcgSolver::new(1e-10, 100).with_monitor(move |i, r| seen2.borrow_mut().push((i, r)));
        let st = s.solve(&a, Some(&pc), &b, &mut x).unwrap();
        assert!(st.iterations <= 10);
        assert!(x.iter().all(|v| (v - 1.0).abs() < 1e-8));
        assert_eq!(seen.borrow().len(), s.residual_history.len());       // live monitor: one callback per history entry, in order
    }

    #[test]
    fn foreign_preconditioner_is_refused() {
        struct Host;
        impl Preconditioner<HipCsrMatrix, Vec<f64>> for Host {
            fn apply(&self, r: &Vec<f64>, z: &mut Vec<f64>) -> Result<(), KError> { for (zi, ri) in z.iter_mut().zip(r) { *zi = *ri; } Ok(()) }
        }
        assert!(probe_device_pc(&Host).is_none());
    }
}
