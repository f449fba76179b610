// kryst_hip.hpp -- header-only C++17 mirror of kryst's operator / preconditioner / solver interface over the C ABI
// of kryst_hip.h.  The reference is a compiled (Rust) crate whose toolchain is absent from the build image, so this is
// the compiled-language host side of the drop-in: same type names, constructor arguments, builder methods, public fields
// and error behaviour as the reference (paths relative to the kryst crate):
//
//   trait MatVec<V>            src/core/traits.rs:4-7          -> struct MatVec<V>           (pure virtual matvec)
//   trait Preconditioner<M,V>  src/preconditioner/mod.rs:8-13  -> struct Preconditioner<M,V> (apply / setup)
//   trait LinearSolver<M,V>    src/solver/mod.rs:30-52         -> struct LinearSolver<M,V>   (solve(a, pc, b, x) -> SolveStats)
//   CsrMatrix::from_csr        src/matrix/sparse.rs:28-46      -> HipCsrMatrix::from_csr
//   Jacobi / Ilu0 / Ilup / Chebyshev / apply_chebyshev          src/preconditioner/*.rs
//   CgSolver / PcgSolver / GmresSolver / FgmresSolver / BiCgStabSolver / CgsSolver / TfqmrSolver   src/solver/*.rs (new(..), with_norm, with_monitor, ...)
//   Convergence, SolveStats    src/utils/convergence.rs:4-14 ;  KError  src/error.rs:6-19 (thrown where Rust returns Err)
//
// V is std::vector<double> (the reference's Vec<f64>).  `Result<T, KError>` becomes "return T or throw KError";
// Rust's assert_eq! panics on length mismatches become KError{ArgumentError}.
#pragma once
#include <functional>
#include <memory>
#include <optional>
#include <stdexcept>
#include <string>
#include <vector>
#include "kryst_hip.h"

namespace kryst {

using Vec = std::vector<double>;

struct KError : std::runtime_error {            // src/error.rs:6-19
    enum Kind { FactorError = 1, SolveError = 2, IndefiniteMatrix = 3, IndefinitePreconditioner = 4, ZeroPivot = 5,
                Unsupported = 6, HipError = 100, RcclError = 101, ArgumentError = 102, CsrError = 103, ContextBusy = 104 };
    int code;
    long long row;                              // KError::ZeroPivot(row), error.rs:15-16 (-1 for every other kind)
    explicit KError(int c) : std::runtime_error(std::string("kryst: ") + kind_name(c) + ": " + kryst_hip_last_error()), code(c),
                             row(c == KRYST_ZERO_PIVOT ? (long long)kryst_hip_last_error_row() : -1) {}
    static const char* kind_name(int c) {
        switch (c) { case 1: return "FactorError"; case 2: return "SolveError"; case 3: return "IndefiniteMatrix";
                     case 4: return "IndefinitePreconditioner"; case 5: return "ZeroPivot"; case 6: return "Unsupported";
                     case 100: return "HipError"; case 101: return "RcclError"; case 102: return "ArgumentError";
                     case 103: return "CsrError"; case 104: return "ContextBusy"; default: return "Error"; }
    }
};
inline void check(int32_t rc) { if (rc != KRYST_OK) throw KError(rc); }

template <class T> struct Convergence { T tol; size_t max_iters; };                          // convergence.rs:4-7
template <class T> struct SolveStats { size_t iterations; T final_residual; bool converged; };   // convergence.rs:10-14

template <class V> struct MatVec { virtual ~MatVec() = default; virtual void matvec(const V& x, V& y) const = 0; };
template <class M, class V> struct Preconditioner {
    virtual ~Preconditioner() = default;
    virtual void apply(const V& r, V& z) const = 0;          // Err(KError) -> throw
    virtual void setup(const M&) {}
    virtual kryst_pc_t device_handle() const { return nullptr; }   // additive hook
};
template <class M, class V> struct LinearSolver {
    virtual ~LinearSolver() = default;
    virtual SolveStats<double> solve(const M& a, const Preconditioner<M, V>* pc, const V& b, V& x) = 0;
};

// One GPU.  Replaces RayonComm / MpiComm (src/parallel).
class Context {
public:
    explicit Context(int device = 0) { check(kryst_ctx_create(device, &h_)); }
    Context(int device, int rank, int nranks, const void* unique_id128) { check(kryst_ctx_create_dist(device, rank, nranks, unique_id128, &h_)); }
    ~Context() { kryst_ctx_destroy(h_); }
    Context(const Context&) = delete; Context& operator=(const Context&) = delete;
    kryst_ctx_t handle() const { return h_; }
    // DistributedInnerProduct (core/wrappers.rs:134-156) inside the solvers: false = RCCL all-gather + rank-ordered fold (default),
    // true = hipIpc mailboxes (one launch, no collective, the same bits).  Collective.  Returns whether the mailbox path is in use.
    bool scalar_reduce_ipc(bool on) {
        int32_t active = 0;
        const int32_t rc = kryst_ctx_scalar_reduce(h_, on ? 1 : 0, &active);
        if (rc != KRYST_OK && rc != KRYST_UNSUPPORTED) check(rc);
        return active != 0;
    }
    static std::shared_ptr<Context> global() { static std::shared_ptr<Context> c = std::make_shared<Context>(0); return c; }
private:
    kryst_ctx_t h_ = nullptr;
};

// CsrMatrix<f64> resident in HBM; SparseMatrix::{nrows,ncols,spmv} (sparse.rs:4-11,49-68) and MatVec.
class HipCsrMatrix : public MatVec<Vec> {
public:
    static HipCsrMatrix from_csr(size_t nrows, size_t ncols, const std::vector<size_t>& row_ptr, const std::vector<size_t>& col_idx,
                                 const Vec& values, std::shared_ptr<Context> ctx = Context::global()) {
        static_assert(sizeof(size_t) == sizeof(uint64_t), "usize is 64-bit");
        if (row_ptr.size() != nrows + 1 || col_idx.size() != values.size()) throw KError(KRYST_ERR_ARG);
        kryst_csr_t h = nullptr;
        check(kryst_csr_create(ctx->handle(), (int64_t)nrows, (int64_t)ncols, reinterpret_cast<const uint64_t*>(row_ptr.data()),
                               reinterpret_cast<const uint64_t*>(col_idx.data()), values.data(), &h));
        return HipCsrMatrix(std::move(ctx), h, nrows, ncols);
    }
    static HipCsrMatrix stencil7(int N, int kind, std::shared_ptr<Context> ctx = Context::global()) {
        kryst_csr_t h = nullptr;
        check(kryst_csr_create_stencil7(ctx->handle(), N, kind, &h));
        const size_t n = (size_t)N * N * N;
        return HipCsrMatrix(std::move(ctx), h, n, n);
    }
    HipCsrMatrix(HipCsrMatrix&& o) noexcept : ctx_(std::move(o.ctx_)), h_(o.h_), nrows_(o.nrows_), ncols_(o.ncols_) { o.h_ = nullptr; }
    ~HipCsrMatrix() override { if (h_) kryst_csr_destroy(h_); }
    size_t nrows() const { return nrows_; }
    size_t ncols() const { return ncols_; }
    void spmv(const Vec& x, Vec& y) const {                   // sparse.rs:56-67 (asserts -> ArgumentError)
        check(kryst_spmv_host(h_, x.data(), (int64_t)x.size(), y.data(), (int64_t)y.size()));
    }
    void matvec(const Vec& x, Vec& y) const override { spmv(x, y); }
    // The halo exchange of a row-partitioned operator (the neighbour exchange src/parallel/mpi_comm.rs:133-143 leaves as a TODO): false =
    // grouped ncclSend / ncclRecv (default), true = direct peer stores into hipIpc-mapped landing buffers (no collective launch, the same
    // bits).  Collective.  Returns whether the peer-store path is in use (it is not when a rank cannot map a peer's buffer).
    bool halo_peer_stores(bool on) {
        int32_t active = 0;
        const int32_t rc = kryst_csr_halo_mode(h_, on ? 1 : 0, &active);
        if (rc != KRYST_OK && rc != KRYST_UNSUPPORTED) check(rc);
        return active != 0;
    }
    kryst_csr_t handle() const { return h_; }
    const std::shared_ptr<Context>& context() const { return ctx_; }
private:
    HipCsrMatrix(std::shared_ptr<Context> c, kryst_csr_t h, size_t nr, size_t nc) : ctx_(std::move(c)), h_(h), nrows_(nr), ncols_(nc) {}
    std::shared_ptr<Context> ctx_; kryst_csr_t h_; size_t nrows_, ncols_;
};

// ---- preconditioners --------------------------------------------------------------------------------------------
class DevicePc : public Preconditioner<HipCsrMatrix, Vec> {
public:
    ~DevicePc() override { if (h_) kryst_pc_destroy(h_); }
    void apply(const Vec& r, Vec& z) const override {
        if (!h_ || !ctx_) throw KError(KRYST_SOLVE_ERROR);
        if (r.size() != z.size()) throw KError(KRYST_ERR_ARG);
        kryst_vec_t rv = nullptr, zv = nullptr;
        check(kryst_vec_create(ctx_, (int64_t)r.size(), &rv));
        int32_t rc = kryst_vec_create(ctx_, (int64_t)z.size(), &zv);
        if (rc == 0) rc = kryst_vec_upload(rv, r.data(), (int64_t)r.size());
        if (rc == 0) rc = kryst_pc_apply(h_, rv, zv);
        if (rc == 0) rc = kryst_vec_download(zv, z.data(), (int64_t)z.size());
        kryst_vec_destroy(rv); kryst_vec_destroy(zv);
        check(rc);
    }
    kryst_pc_t device_handle() const override { return h_; }
protected:
    void reset(kryst_pc_t h, kryst_ctx_t c) { if (h_) kryst_pc_destroy(h_); h_ = h; ctx_ = c; }
    kryst_pc_t h_ = nullptr; kryst_ctx_t ctx_ = nullptr;
};
struct Jacobi : DevicePc {                                   // jacobi.rs:26-95
    void setup(const HipCsrMatrix& a) override { kryst_pc_t h = nullptr; check(kryst_pc_jacobi(a.handle(), &h)); reset(h, a.context()->handle()); }
};
struct Ilu0 : DevicePc {                                     // ilu.rs:32-122 (as written)
    void setup(const HipCsrMatrix& a) override { kryst_pc_t h = nullptr; check(kryst_pc_ilu0(a.handle(), KRYST_ILU_KRYST_COMPAT, &h)); reset(h, a.context()->handle()); }
};
struct Ilup : DevicePc {                                     // ilup.rs:54-167 (level-of-fill p, as written)
    explicit Ilup(size_t fill = 0) : fill(fill) {}
    size_t fill;
    void setup(const HipCsrMatrix& a) override { kryst_pc_t h = nullptr; check(kryst_pc_ilup(a.handle(), (int32_t)fill, &h)); reset(h, a.context()->handle()); }
};
struct Ilut : DevicePc {                                     // ilut.rs:55-150 (as written)
    Ilut(size_t fill, double droptol) : fill(fill), droptol(droptol) {}
    size_t fill; double droptol;
    void setup(const HipCsrMatrix& a) override { kryst_pc_t h = nullptr; check(kryst_pc_ilut(a.handle(), (int32_t)fill, droptol, &h)); reset(h, a.context()->handle()); }
};
struct TrueIlu0 : DevicePc {                                 // extension
    void setup(const HipCsrMatrix& a) override { kryst_pc_t h = nullptr; check(kryst_pc_ilu0(a.handle(), KRYST_ILU_TRUE_ILU0, &h)); reset(h, a.context()->handle()); }
};
struct IdentityPC : DevicePc {                               // pcg.rs:245-251
    void setup(const HipCsrMatrix& a) override { kryst_pc_t h = nullptr; check(kryst_pc_identity(a.context()->handle(), &h)); reset(h, a.context()->handle()); }
};
// ApproxInv with GIVEN inverse rows (ApproxInv::inv_rows, approxinv.rs:66): apply (approxinv.rs:268-298) is z = M r on the device.
// ApproxInv::setup (least squares through faer's QR) stays with the reference; setup() here only checks the size.
struct ApproxInv : DevicePc {
    explicit ApproxInv(const std::vector<std::vector<std::pair<size_t, double>>>& inv_rows, std::shared_ptr<Context> ctx = Context::global())
        : m_(build(inv_rows, std::move(ctx))) {
        kryst_pc_t h = nullptr; check(kryst_pc_approx_inverse(m_.handle(), &h)); reset(h, m_.context()->handle());
    }
    void setup(const HipCsrMatrix& a) override { if (a.nrows() != m_.nrows()) throw KError(KRYST_ERR_ARG); }
private:
    static HipCsrMatrix build(const std::vector<std::vector<std::pair<size_t, double>>>& rows, std::shared_ptr<Context> ctx) {
        std::vector<size_t> rp(rows.size() + 1, 0), ci; Vec va;
        for (size_t i = 0; i < rows.size(); ++i) { for (auto& e : rows[i]) { ci.push_back(e.first); va.push_back(e.second); } rp[i + 1] = ci.size(); }
        return HipCsrMatrix::from_csr(rows.size(), rows.size(), rp, ci, va, std::move(ctx));
    }
    HipCsrMatrix m_;
};
struct Chebyshev : DevicePc {                                // chebyshev.rs:35-70: the trait apply is a stub returning Err
    size_t degree; std::optional<double> lambda_min, lambda_max;
    Chebyshev(size_t degree, std::optional<double> lmin, std::optional<double> lmax) : degree(degree), lambda_min(lmin), lambda_max(lmax) {}
    void setup(const HipCsrMatrix& a) override { kryst_pc_t h = nullptr; check(kryst_pc_chebyshev_stub(a.context()->handle(), (int32_t)degree, &h)); reset(h, a.context()->handle()); }
};
inline void apply_chebyshev(const HipCsrMatrix& a, const Vec& r, Vec& z, double alpha, double beta, size_t m) {   // chebyshev.rs:83-140
    kryst_ctx_t c = a.context()->handle();
    kryst_vec_t rv = nullptr, zv = nullptr;
    check(kryst_vec_create(c, (int64_t)r.size(), &rv));
    int32_t rc = kryst_vec_create(c, (int64_t)z.size(), &zv);
    if (rc == 0) rc = kryst_vec_upload(rv, r.data(), (int64_t)r.size());
    if (rc == 0) rc = kryst_apply_chebyshev(a.handle(), rv, zv, alpha, beta, (int64_t)m);
    if (rc == 0) rc = kryst_vec_download(zv, z.data(), (int64_t)z.size());
    kryst_vec_destroy(rv); kryst_vec_destroy(zv);
    check(rc);
}

// ---- solvers -------------------------------------------------------------------------------------------------------
enum class CgNormType { Preconditioned = 0, Unpreconditioned = 1, Natural = 2, None = 3 };      // cg.rs:35
enum class Preconditioning { None = 0, Left = 1, Right = 2, LeftTextbook = 3 };                  // gmres.rs:28-32; LeftTextbook: labelled extension (kryst_hip.h: precond_side 3)

class SolverBase : public LinearSolver<HipCsrMatrix, Vec> {
public:
    Convergence<double> conv;
    CgNormType norm_type = CgNormType::Unpreconditioned;
    bool single_reduction = false;
    std::optional<double> radius, obj_target;
    std::function<void(size_t, double)> monitor;     // with_monitor (cg.rs:84-88): fired live, in order, on the calling thread
    std::vector<double> residual_history;
    int check_every = 0;                             // iterations between two rounds of monitor callbacks (0: the library's 8)
    void clear_history() { residual_history.clear(); }
    SolveStats<double> solve(const HipCsrMatrix& a, const Preconditioner<HipCsrMatrix, Vec>* pc, const Vec& b, Vec& x) override {
        if (b.size() != x.size()) throw KError(KRYST_ERR_ARG);
        kryst_params_t p{};
        p.tol = conv.tol; p.max_iters = (int64_t)conv.max_iters; p.restart = restart_; p.precond_side = side_;
        p.norm_type = (int)norm_type; p.single_reduction = single_reduction;
        p.has_radius = radius.has_value(); p.radius = radius.value_or(0.0);
        p.has_obj_target = obj_target.has_value(); p.obj_target = obj_target.value_or(0.0);
        p.check_every = check_every;
        kryst_stats_t st{};
        std::vector<double> hist(std::min<size_t>((size_t)hist_per_iter_ * conv.max_iters + (size_t)(restart_ > 0 ? restart_ : 1) + 8,
                                                  ((size_t)1 << 22) + 8));      // the library records at most 2^22 entries
        int64_t len = 0;
        const int32_t rc = call(b.data(), x.data(), (int64_t)b.size(), a.handle(), pc ? pc->device_handle() : nullptr, &p, &st,
                                hist.data(), (int64_t)hist.size(), &len, monitor ? &SolverBase::trampoline : nullptr, this);
        const size_t k = (size_t)std::min<int64_t>(len, (int64_t)hist.size());
        residual_history.insert(residual_history.end(), hist.begin(), hist.begin() + (long)k);
        check(rc);
        return SolveStats<double>{(size_t)st.iterations, st.final_residual, st.converged != 0};
    }
protected:
    SolverBase(double tol, size_t max_iters) : conv{tol, max_iters} {}
    virtual int32_t call(const double* b, double* x, int64_t n, KRYST_SOLVE_ARGS) = 0;
    int restart_ = 0, side_ = 1, hist_per_iter_ = 1;
private:
    static void trampoline(int64_t it, double res, void* user) { static_cast<SolverBase*>(user)->monitor((size_t)it, res); }
};
#define KRYST_FWD a, pc, params, stats, hist, hist_cap, hist_len, monitor, user

struct CgSolver : SolverBase {                               // cg.rs:40-93
    CgSolver(double tol, size_t max_iters) : SolverBase(tol, max_iters) {}
    static CgSolver create(double tol, size_t max_iters) { return CgSolver(tol, max_iters); }     // CgSolver::new
    CgSolver& with_norm(CgNormType t) { norm_type = t; return *this; }
    CgSolver& with_single_reduction(bool f) { single_reduction = f; return *this; }
    CgSolver& with_radius(double r) { radius = r; return *this; }
    CgSolver& with_obj_target(double o) { obj_target = o; return *this; }
    CgSolver& with_monitor(std::function<void(size_t, double)> f) { monitor = std::move(f); return *this; }
protected:
    int32_t call(const double* b, double* x, int64_t n, KRYST_SOLVE_ARGS) override { return kryst_cg_solve(b, x, n, KRYST_FWD); }
};
struct PcgSolver : SolverBase {                              // pcg.rs:31-91
    PcgSolver(double tol, size_t max_iters) : SolverBase(tol, max_iters) {}
    PcgSolver& with_norm(CgNormType t) { norm_type = t; return *this; }
    PcgSolver& with_single_reduction(bool f) { single_reduction = f; return *this; }
    PcgSolver& with_monitor(std::function<void(size_t, double)> f) { monitor = std::move(f); return *this; }
protected:
    int32_t call(const double* b, double* x, int64_t n, KRYST_SOLVE_ARGS) override { return kryst_pcg_solve(b, x, n, KRYST_FWD); }
};
struct GmresSolver : SolverBase {                            // gmres.rs:38-60
    size_t restart; Preconditioning preconditioning = Preconditioning::Left;
    GmresSolver(size_t restart, double tol, size_t max_iters) : SolverBase(tol, max_iters), restart(restart) { restart_ = (int)restart; }
    GmresSolver& with_preconditioning(Preconditioning m) { preconditioning = m; side_ = (int)m; return *this; }
protected:
    int32_t call(const double* b, double* x, int64_t n, KRYST_SOLVE_ARGS) override { return kryst_gmres_solve(b, x, n, KRYST_FWD); }
};
enum class Orthog { Classical = 0, Modified = 1 };           // fgmres.rs:26-31
struct FgmresSolver : SolverBase {                           // fgmres.rs:33-101; solve_flex :114-340
    size_t restart; Orthog orthog = Orthog::Classical; double haptol = 1e-12; bool preallocate = false; size_t delta_allocate = 10;
    FgmresSolver(double tol, size_t max_iters, size_t restart) : SolverBase(tol, max_iters), restart(restart) { restart_ = (int)restart; }
    FgmresSolver& with_orthog(Orthog o) { orthog = o; return *this; }
    FgmresSolver& with_preallocate(bool f) { preallocate = f; return *this; }
    FgmresSolver& with_delta_allocate(size_t d) { delta_allocate = d; return *this; }
    FgmresSolver& with_haptol(double h) { haptol = h; return *this; }
    // the FlexiblePreconditioner (preconditioner/mod.rs:16-19) is a device preconditioner object
    SolveStats<double> solve_flex(const HipCsrMatrix& a, const Preconditioner<HipCsrMatrix, Vec>* pc, const Vec& b, Vec& x) { return solve(a, pc, b, x); }
    FgmresSolver& with_monitor(std::function<void(size_t, double)> f) { monitor = std::move(f); return *this; }
protected:
    int32_t call(const double* b, double* x, int64_t n, KRYST_SOLVE_ARGS) override {
        return kryst_fgmres_solve(b, x, n, (int32_t)orthog, haptol, preallocate ? 1 : 0, KRYST_FWD);
    }
};
struct BiCgStabSolver : SolverBase {                         // bicgstab.rs:36-48
    BiCgStabSolver(double tol, size_t max_iters) : SolverBase(tol, max_iters) {}
protected:
    int32_t call(const double* b, double* x, int64_t n, KRYST_SOLVE_ARGS) override { return kryst_bicgstab_solve(b, x, n, KRYST_FWD); }
};
struct CgsSolver : SolverBase {                              // cgs.rs:21-35
    CgsSolver(double tol, size_t max_iters) : SolverBase(tol, max_iters) {}
protected:
    int32_t call(const double* b, double* x, int64_t n, KRYST_SOLVE_ARGS) override { return kryst_cgs_solve(b, x, n, KRYST_FWD); }
};
struct TfqmrSolver : SolverBase {                            // tfqmr.rs:30-40
    TfqmrSolver(double tol, size_t max_iters) : SolverBase(tol, max_iters) { hist_per_iter_ = 2; }
protected:
    int32_t call(const double* b, double* x, int64_t n, KRYST_SOLVE_ARGS) override { return kryst_tfqmr_solve(b, x, n, KRYST_FWD); }
};
#undef KRYST_FWD

// ---- context/: PC<T> (src/context/pc_context.rs:36-76) and KspContext (src/context/ksp_context.rs:25-148) ---------------------
// PC<T>: the reference's configuration enum for preconditioners, plus the constructor it lacks -- build(a) returns the set-up
// device preconditioner.  Kinds outside the hot path (Ssor, ApproxInv setup, BlockJacobi, Multicolor, AMG, AdditiveSchwarz)
// throw KError{Unsupported}.
struct PC {
    enum Kind { JacobiKind, SsorKind, Ilu0Kind, IlupKind, IlutKind, ChebyshevKind, ApproxInvKind, BlockJacobiKind, MulticolorKind, AMGKind, AdditiveSchwarzKind };
    Kind kind; size_t fill = 0; double droptol = 0.0; size_t degree = 0; std::optional<double> emin, emax;
    static PC Jacobi() { return PC{JacobiKind}; }
    static PC Ilu0() { return PC{Ilu0Kind}; }
    static PC Ilup(size_t fill) { PC p{IlupKind}; p.fill = fill; return p; }
    static PC Ilut(size_t fill, double droptol) { PC p{IlutKind}; p.fill = fill; p.droptol = droptol; return p; }
    static PC Chebyshev(size_t degree, std::optional<double> emin = std::nullopt, std::optional<double> emax = std::nullopt) {
        PC p{ChebyshevKind}; p.degree = degree; p.emin = emin; p.emax = emax; return p;
    }
    std::unique_ptr<Preconditioner<HipCsrMatrix, Vec>> build(const HipCsrMatrix& a) const {
        std::unique_ptr<Preconditioner<HipCsrMatrix, Vec>> pc;
        switch (kind) {
            case JacobiKind: pc = std::make_unique<kryst::Jacobi>(); break;
            case Ilu0Kind: pc = std::make_unique<kryst::Ilu0>(); break;
            case IlupKind: pc = std::make_unique<kryst::Ilup>(fill); break;
            case IlutKind: pc = std::make_unique<kryst::Ilut>(fill, droptol); break;
            case ChebyshevKind: pc = std::make_unique<kryst::Chebyshev>(degree, emin, emax); break;     // the trait object: apply is the stub
            default: throw KError(KRYST_UNSUPPORTED);
        }
        pc->setup(a);
        return pc;
    }
};

enum class SolverKind { Cg, Pcg, GmresLeft, GmresRight, Fgmres, Bicgstab, Cgs, Qmr, Tfqmr, Minres, Cgnr };   // ksp_context.rs:25-50

// KspContext { kind, a, pc, flex_pc, tol, max_it, restart } + solve_context (ksp_context.rs:54-148): a fresh solver of `kind` per
// call, forwarded (a, pc, b, x) exactly as the reference's match does -- FGMRES uses flex_pc, never pc (:101-107); the kinds that
// need A^T or are outside the accelerated path (Qmr, Minres, Cgnr) throw KError{Unsupported}.  `a` is borrowed (the reference
// owns an M by value; a device operator is not copyable).
struct KspContext {
    SolverKind kind;
    const HipCsrMatrix& a;
    std::unique_ptr<Preconditioner<HipCsrMatrix, Vec>> pc;
    std::unique_ptr<Preconditioner<HipCsrMatrix, Vec>> flex_pc;
    double tol; size_t max_it; size_t restart;
    KspContext(SolverKind kind, const HipCsrMatrix& a, std::unique_ptr<Preconditioner<HipCsrMatrix, Vec>> pc, double tol, size_t max_it,
               size_t restart = 30, std::unique_ptr<Preconditioner<HipCsrMatrix, Vec>> flex_pc = nullptr)
        : kind(kind), a(a), pc(std::move(pc)), flex_pc(std::move(flex_pc)), tol(tol), max_it(max_it), restart(restart) {}
    SolveStats<double> solve_context(const Vec& b, Vec& x) {
        switch (kind) {
            case SolverKind::GmresLeft: { GmresSolver s(restart, tol, max_it); s.with_preconditioning(Preconditioning::Left); return s.solve(a, pc.get(), b, x); }
            case SolverKind::GmresRight: { GmresSolver s(restart, tol, max_it); s.with_preconditioning(Preconditioning::Right); return s.solve(a, pc.get(), b, x); }
            case SolverKind::Fgmres: { FgmresSolver s(tol, max_it, restart); return s.solve_flex(a, flex_pc.get(), b, x); }
            case SolverKind::Cg: { CgSolver s(tol, max_it); return s.solve(a, pc.get(), b, x); }
            case SolverKind::Pcg: { PcgSolver s(tol, max_it); return s.solve(a, pc.get(), b, x); }
            case SolverKind::Bicgstab: { BiCgStabSolver s(tol, max_it); return s.solve(a, pc.get(), b, x); }
            case SolverKind::Cgs: { CgsSolver s(tol, max_it); return s.solve(a, pc.get(), b, x); }
            case SolverKind::Tfqmr: { TfqmrSolver s(tol, max_it); return s.solve(a, pc.get(), b, x); }
            default: throw KError(KRYST_UNSUPPORTED);
        }
    }
};

}  // namespace kryst
