/*
 * kryst_oracle.h -- CPU ORACLE for the kryst Krylov hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This is a plain-C restatement, operation by operation, of the reference crate
 * tmathis720/kryst v0.5.3 (paths below are relative to /root/reference):
 *   src/core/wrappers.rs:27-38,90-127      dense row loop order, dot / norm
 *   src/matrix/sparse.rs:103-114           spmv_parallel row sum (ascending column, start 0, no FMA)
 *   src/solver/cg.rs:114-288               CgSolver::solve
 *   src/solver/pcg.rs:114-222              PcgSolver::solve
 *   src/solver/gmres.rs:65-105,154-192,216-402   GmresSolver::solve (+arnoldi, givens, back_substitution)
 *   src/solver/bicgstab.rs:69-293          BiCgStabSolver::solve
 *   src/preconditioner/jacobi.rs:53-95     Jacobi setup/apply
 *   src/preconditioner/chebyshev.rs:83-159 apply_chebyshev / chebyshev_t
 *   src/preconditioner/ilu.rs:59-122       Ilu0 (dense "ILU" as written: L=I+tril(A,-1)D^-1, U=I+triu(A,1))
 *   src/preconditioner/ilup.rs:77-167      Ilup::new(0) (as written: no elimination at p=0 => L same, U=triu(A))
 *   src/utils/convergence.rs:18-34         Convergence::check
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use this code, and only as
 * the checker / the reported CPU baseline.  The product (kryst_amd/, libkryst_hip.so) never links,
 * imports or calls it.
 *
 * PARITY PINNING: the reference is Rust + un-vendored crates.io dependencies (faer 0.22.6 etc.);
 * no cargo/rustc exists in the build container, so the reference itself cannot be compiled or run
 * (no oracle/_ref).  The oracle is pinned by the reference's own known-answer tests
 * (tests/test_oracle_reference_pins.py re-encodes every one of them, file:line cited there).
 *
 * Arithmetic contract: IEEE fp64, compiled with -ffp-contract=off (Rust never contracts a*b+c).
 *
 * Inner products come in two association orders:
 *   KRO_REDUCE_SERIAL  strict left fold  == the reference built with --no-default-features
 *                      (wrappers.rs:101-107,120-126); the canonical, deterministic definition.
 *   KRO_REDUCE_TILED   the fixed tree the HIP kernels use (tile = T*V elements; per-thread fold,
 *                      64-lane xor butterfly, serial across waves; tile partials folded in chunks of F
 *                      -- one per thread, butterfly, serial across waves -- and the chunk values likewise;
 *                      rank partials folded in rank order).
 *                      The default-feature reference uses Rayon's reduce (wrappers.rs:92-100), whose
 *                      association is unspecified and run-dependent, so every fixed tree is one of its
 *                      admissible executions.  This mode lets the GPU be checked BIT-FOR-BIT.
 */
#ifndef KRYST_ORACLE_H
#define KRYST_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { KRO_OK = 0, KRO_FACTOR_ERROR = 1, KRO_SOLVE_ERROR = 2, KRO_INDEFINITE_MATRIX = 3,
       KRO_INDEFINITE_PC = 4, KRO_ZERO_PIVOT = 5, KRO_UNSUPPORTED = 6 };   /* error.rs:6-19 */

enum { KRO_REDUCE_SERIAL = 0, KRO_REDUCE_TILED = 1 };

typedef struct {
    int32_t mode;           /* KRO_REDUCE_* */
    int32_t T;              /* threads per tile (multiple of 64) */
    int32_t V;              /* elements per thread per tile */
    int32_t F;              /* threads of the final fold (multiple of 64) */
    int32_t nparts;         /* rank partition of the vector; 0 or 1 = single part */
    const int64_t* part_off;/* nparts+1 offsets (ignored when nparts <= 1) */
} kro_reduce_t;

typedef struct {
    int64_t nrows, ncols;
    const int64_t* row_ptr; /* nrows+1 */
    const int64_t* col_idx; /* nnz, ascending and duplicate-free inside a row (sparse.rs:36-42 new_checked) */
    const double*  vals;    /* nnz */
} kro_csr_t;

enum { KRO_PC_NONE = 0,       /* pc == None */
       KRO_PC_IDENTITY = 1,   /* test IdentityPC, pcg.rs:245-251 */
       KRO_PC_JACOBI = 2,     /* jacobi.rs */
       KRO_PC_ILU0_COMPAT = 3,/* ilu.rs as written */
       KRO_PC_ILUP0 = 4,      /* ilup.rs with fill = 0 as written */
       KRO_PC_ILU0_TRUE = 5,  /* extension: textbook IKJ ILU(0) on A's pattern (Saad Alg. 10.4) */
       KRO_PC_CHEBYSHEV_STUB = 6, /* chebyshev.rs:68-70: apply returns Err(SolveError) */
       KRO_PC_CHEBYSHEV = 7,  /* extension: apply == apply_chebyshev(a, r, z, alpha, beta, m) */
       KRO_PC_TRIROWS = 8,    /* Ilup(p) / Ilut factors as explicit sparse rows (ilup.rs:54-59, ilut.rs:55-61) */
       KRO_PC_SPAI = 9        /* ApproxInv::apply with given inv_rows (approxinv.rs:268-298): z = M r, M = pc->a */ };

/* L and U as Vec<SparseRow> (ilup.rs:29-33): per row the (col, val) pairs IN STORED ORDER */
typedef struct {
    int64_t n;
    int64_t* l_ptr; int64_t* l_col; double* l_val;
    int64_t* u_ptr; int64_t* u_col; double* u_val;
} kro_trirows_t;

typedef struct {
    int32_t kind;
    const kro_csr_t* a;     /* matrix the factors refer to (pattern for ILU kinds, operator for Chebyshev) */
    const double* inv_diag; /* JACOBI: n values */
    const double* lfac;     /* ILU kinds: value per nnz of A; meaningful where col < row (strict lower) */
    const double* ufac;     /* ILU kinds: value per nnz of A; meaningful where col >= row */
    int32_t divide_diag;    /* ILU kinds: 1 = back substitution divides by the stored diagonal */
    double cheb_alpha, cheb_beta; int32_t cheb_degree;
    const kro_trirows_t* rows;  /* TRIROWS */
} kro_pc_t;

typedef struct {
    double  tol;
    int64_t max_iters;
    int32_t restart;          /* GMRES */
    int32_t precond_side;     /* GMRES: 0 None, 1 Left (default, gmres.rs:53), 2 Right; 3 = textbook Left (labelled extension, not in the reference) */
    int32_t norm_type;        /* CG/PCG CgNormType: 0 Preconditioned, 1 Unpreconditioned (default), 2 Natural, 3 None */
    int32_t single_reduction; /* cg.rs:146-165, pcg.rs:151-160 */
    int32_t has_radius;   double radius;      /* cg.rs:177-202 */
    int32_t has_obj_target; double obj_target;/* cg.rs:231-252 */
} kro_params_t;

typedef struct { int64_t iterations; double final_residual; int32_t converged; } kro_stats_t; /* convergence.rs:10-14 */

typedef void (*kro_monitor_fn)(int64_t iter, double res, void* user);

typedef struct {            /* optional per-iteration record (cg.rs:46-47 residual_history + monitor) */
    double* hist; int64_t cap; int64_t len;
    kro_monitor_fn monitor; void* user;
} kro_trace_t;

void   kro_set_threads(int32_t nthreads);     /* OpenMP threads for row / tile loops (results do not depend on it) */
int32_t kro_get_threads(void);

double kro_dot (const kro_reduce_t* rs, const double* x, const double* y, int64_t n);
double kro_norm(const kro_reduce_t* rs, const double* x, int64_t n);
void   kro_spmv(const kro_csr_t* a, const double* x, double* y);
int32_t kro_csr_check(const kro_csr_t* a);    /* 0 ok; else violated precondition */

int32_t kro_jacobi_setup(const kro_csr_t* a, double* inv_diag);
int32_t kro_ilu0_compat_setup(const kro_csr_t* a, double* lfac, double* ufac);
int32_t kro_ilup0_setup(const kro_csr_t* a, double* lfac, double* ufac);
int32_t kro_ilu0_true_setup(const kro_csr_t* a, double* lfac, double* ufac);
/* Ilup::setup (ilup.rs:77-134) and Ilut::setup (ilut.rs:80-117) exactly as written, dense n x n work arrays included
 * (small n only); the rows come back in the order the reference pushes them */
int32_t kro_ilup_build(const kro_csr_t* a, int64_t fill, kro_trirows_t* out);
int32_t kro_ilut_build(const kro_csr_t* a, int64_t fill, double droptol, kro_trirows_t* out);
void    kro_trirows_free(kro_trirows_t* t);
int32_t kro_pc_apply(const kro_pc_t* pc, const double* r, double* z, int64_t n);
void   kro_apply_chebyshev(const kro_csr_t* a, const double* r, double* z, int64_t n,
                           double alpha, double beta, int64_t m);
double kro_chebyshev_t(int64_t m, double x);

int32_t kro_cg      (const kro_csr_t* a, const kro_pc_t* pc, const double* b, double* x,
                     const kro_params_t* p, const kro_reduce_t* rs, kro_stats_t* st, kro_trace_t* tr);
int32_t kro_pcg     (const kro_csr_t* a, const kro_pc_t* pc, const double* b, double* x,
                     const kro_params_t* p, const kro_reduce_t* rs, kro_stats_t* st, kro_trace_t* tr);
int32_t kro_gmres   (const kro_csr_t* a, const kro_pc_t* pc, const double* b, double* x,
                     const kro_params_t* p, const kro_reduce_t* rs, kro_stats_t* st, kro_trace_t* tr);
int32_t kro_bicgstab(const kro_csr_t* a, const kro_pc_t* pc, const double* b, double* x,
                     const kro_params_t* p, const kro_reduce_t* rs, kro_stats_t* st, kro_trace_t* tr);
/* CgsSolver::solve (src/solver/cgs.rs:58-135; pc ignored :59) and TfqmrSolver::solve (src/solver/tfqmr.rs:64-221; pc ignored,
 * the initial guess is overwritten with zeros :72).  The reference keeps no history; the trace receives ||r|| per
 * iteration (CGS) and the residual estimate dpest per substep (TFQMR).  TFQMR has no enabled known-answer test in the
 * reference (its only test is #[ignore], tfqmr.rs:241): parity for it rests on the literal restatement. */
int32_t kro_cgs     (const kro_csr_t* a, const kro_pc_t* pc, const double* b, double* x,
                     const kro_params_t* p, const kro_reduce_t* rs, kro_stats_t* st, kro_trace_t* tr);
int32_t kro_tfqmr   (const kro_csr_t* a, const kro_pc_t* pc, const double* b, double* x,
                     const kro_params_t* p, const kro_reduce_t* rs, kro_stats_t* st, kro_trace_t* tr);
/* FgmresSolver::solve_flex (src/solver/fgmres.rs:114-340): p->restart, p->tol, p->max_iters; orthog 0 Classical (default)
 * / 1 Modified; haptol (default 1e-12); preallocate (default 0).  pc plays the FlexiblePreconditioner (mod.rs:16-19). */
int32_t kro_fgmres  (const kro_csr_t* a, const kro_pc_t* pc, const double* b, double* x,
                     const kro_params_t* p, int32_t orthog, double haptol, int32_t preallocate,
                     const kro_reduce_t* rs, kro_stats_t* st, kro_trace_t* tr);
/* extension (not in the reference, which ignores pc at bicgstab.rs:70): right-preconditioned BiCGStab */
int32_t kro_bicgstab_rpc(const kro_csr_t* a, const kro_pc_t* pc, const double* b, double* x,
                     const kro_params_t* p, const kro_reduce_t* rs, kro_stats_t* st, kro_trace_t* tr);

#ifdef __cplusplus
}
#endif
#endif
