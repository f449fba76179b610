/*
 * kryst_hip.h -- C ABI of libkryst_hip.so: the MI355X (gfx950) Krylov inner loop behind kryst's
 * MatVec / Preconditioner / LinearSolver traits.
 *
 * Every entry point names the reference interface it replaces (paths relative to the kryst crate,
 * tmathis720/kryst v0.5.3).  A Rust maintainer binds these with `extern "C"` (INTEGRATION.md shows the
 * stub); tests and bench.py bind them with ctypes (kryst_amd/_ffi.py).
 *
 * Conventions
 *   - every function returns an int32 status: 0 OK, 1..6 mirror `KError` (src/error.rs:6-19),
 *     >= 100 are HIP / RCCL / argument errors; kryst_hip_last_error() gives the text.  Nothing unwinds
 *     across the ABI.
 *   - handles are opaque, created and destroyed by the library; host arrays are borrowed only for the
 *     duration of a call; one host thread per context; contexts are independent.
 *   - all arithmetic is IEEE fp64 with no FMA contraction; row sums run over ascending stored columns.
 *   - inner products use ONE fixed association tree (kryst_reduce_spec): results are run-to-run and
 *     launch-configuration independent; with nranks > 1 rank results are folded in rank order.
 *   - there is no CPU fallback: without a GPU every compute call fails with KRYST_ERR_HIP.
 *   - Contexts: the scalar state of a solve (alpha, beta, the convergence record, the progress record the host polls)
 *     lives in per-context device scratch, so ONE solve or stepping session can be open per context at a time.  While
 *     a session is open, kryst_*_solve[_dev] and kryst_session_begin on the same context return KRYST_ERR_BUSY;
 *     kryst_spmv, kryst_dot / kryst_norm, the vector updates and kryst_pc_apply stay legal (they are stream-ordered
 *     behind the session's enqueued iterations and use scratch of their own).  Use a second context for a second
 *     concurrent solve.
 */
#ifndef KRYST_HIP_H
#define KRYST_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- status codes: src/error.rs:6-19 ---- */
enum {
    KRYST_OK = 0,
    KRYST_FACTOR_ERROR = 1,             /* KError::FactorError */
    KRYST_SOLVE_ERROR = 2,              /* KError::SolveError */
    KRYST_INDEFINITE_MATRIX = 3,        /* KError::IndefiniteMatrix            cg.rs:168-174, pcg.rs:162-172 */
    KRYST_INDEFINITE_PRECONDITIONER = 4,/* KError::IndefinitePreconditioner    cg.rs:254-259, pcg.rs:208-213 */
    KRYST_ZERO_PIVOT = 5,               /* KError::ZeroPivot(row) */
    KRYST_UNSUPPORTED = 6,              /* KError::Unsupported */
    KRYST_ERR_HIP = 100,                /* HIP runtime error / no device */
    KRYST_ERR_RCCL = 101,               /* RCCL error / library not found */
    KRYST_ERR_ARG = 102,                /* bad argument (length mismatch = the reference's assert_eq! panics) */
    KRYST_ERR_CSR = 103,                /* CSR violates new_checked preconditions (sparse.rs:36-42) */
    KRYST_ERR_BUSY = 104                /* a solve / stepping session is already open on this context (see "Contexts") */
};

typedef struct kryst_ctx_s* kryst_ctx_t;
typedef struct kryst_csr_s* kryst_csr_t;
typedef struct kryst_vec_s* kryst_vec_t;
typedef struct kryst_pc_s*  kryst_pc_t;

const char* kryst_hip_last_error(void);
/* the row of the last KRYST_ZERO_PIVOT on this thread (KError::ZeroPivot(row), src/error.rs:15-16), -1 if none yet */
int64_t     kryst_hip_last_error_row(void);
int32_t     kryst_hip_abi_version(void);
/* The fixed inner-product tree: tile = T*V elements; thread t folds its V elements, 64-lane xor butterfly,
 * serial across the T/64 waves; the tile partials are folded in chunks of F (one per thread, butterfly, serial
 * across the F/64 waves) and, when there is more than one chunk, the chunk values by F threads (stride F) likewise. */
void        kryst_reduce_spec(int32_t* T, int32_t* V, int32_t* F);

/* ---- context: one per GPU / per rank.  Replaces src/parallel (Comm trait, parallel/mod.rs:4-35) ---- */
/* HIP devices this process sees (0 without a GPU): a launcher checks it before it hands LOCAL_RANK to kryst_ctx_create_dist -- one rank
 * per GPU, as MpiComm::new gets one rank per process from mpirun (src/parallel/mpi_comm.rs:49-55) */
int32_t kryst_device_count(int32_t* count);
int32_t kryst_ctx_create(int32_t device_id, kryst_ctx_t* out);
/* rank/nranks + 128-byte RCCL unique id (rank 0 makes it with kryst_comm_unique_id and ships it to the other
 * ranks by any side channel).  Replaces MpiComm::new (src/parallel/mpi_comm.rs:49-55). */
int32_t kryst_comm_unique_id(void* out128);
int32_t kryst_ctx_create_dist(int32_t device_id, int32_t rank, int32_t nranks, const void* unique_id128,
                              kryst_ctx_t* out);
int32_t kryst_ctx_destroy(kryst_ctx_t ctx);
int32_t kryst_ctx_synchronize(kryst_ctx_t ctx);                    /* hipStreamSynchronize on the ctx streams */
int32_t kryst_ctx_rank(kryst_ctx_t ctx, int32_t* rank, int32_t* nranks);   /* Comm::rank / Comm::size */
int32_t kryst_comm_barrier(kryst_ctx_t ctx);                       /* Comm::barrier, mpi_comm.rs:67 */
int32_t kryst_comm_all_reduce(kryst_ctx_t ctx, double x, double* out);     /* Comm::all_reduce, mpi_comm.rs:116-121 */
/* How the solvers' inner products cross the ranks (DistributedInnerProduct, core/wrappers.rs:134-156): mode 0 = RCCL all-gather +
 * rank-ordered fold, 1 = hipIpc-mapped mailboxes written and polled by the kernel that finishes the local fold (one launch, no
 * collective; the same bits), -1 = query (*active only; not collective).  Modes 0 / 1: COLLECTIVE over the context's ranks, no solve
 * open.  Mode 1 returns KRYST_UNSUPPORTED -- on every rank, which all stay on RCCL -- when a mailbox cannot be exported or mapped or ONE
 * CHECKED TEST REDUCTION over the fresh mailboxes does not arrive intact on some rank.  *active (may be NULL): mode in use.
 * DEFAULT (ABI 5): kryst_ctx_create_dist with more than one rank tries mode 1 and keeps it when it works on every rank;
 * KRYST_SCALAR_REDUCE=rccl keeps mode 0. */
int32_t kryst_ctx_scalar_reduce(kryst_ctx_t ctx, int32_t mode, int32_t* active);
/* How a row-partitioned operator's halo exchange travels (the neighbour exchange src/parallel/mpi_comm.rs:133-143 leaves as a TODO): mode 0 =
 * grouped ncclSend / ncclRecv on the second stream, 1 = direct peer stores -- a push kernel writes the rows each neighbour needs
 * straight into that neighbour's hipIpc-mapped landing buffer and stamps the exchange's epoch behind them, the receiver's compute stream
 * polls its stamps in front of the boundary tiles: no collective launch, no pack kernel, no event between receive and compute stream --
 * -1 = query (*active only; not collective).  Modes 0 / 1: COLLECTIVE over the context's ranks, no solve open.  Mode 1 returns
 * KRYST_UNSUPPORTED -- on every rank, which all stay on RCCL -- when a landing buffer cannot be exported or mapped, a neighbour relation is
 * one-way, or ONE CHECKED TEST EXCHANGE (every rank sends its global row numbers and compares what lands with its column list) does not
 * arrive intact on some rank.  The same bits either way.  *active (may be NULL): mode in use.
 * DEFAULT (ABI 5): kryst_csr_create_dist / kryst_csr_create_stencil7 on a context of several ranks try mode 1 and keep it when it works on
 * every rank; KRYST_HALO_MODE=rccl keeps mode 0.  kryst_spmv on an operator in mode 1 returns KRYST_ERR_RCCL when a neighbour's stamp
 * never arrived (the halo was NaNs). */
int32_t kryst_csr_halo_mode(kryst_csr_t a, int32_t mode, int32_t* active);
/* Returns to the driver what the context keeps between calls: the device blocks of destroyed ILU-family preconditioners -- kept, keyed by
 * size, so that the next Preconditioner::setup of the same matrix (ilup.rs:77-134 is called per matrix, repeatedly) costs the factorisation
 * and not 22 GB of allocation at 512^3; bounded by KRYST_DEV_POOL_MB (default 65536, 0 = no pool) -- and, when no solve is open, the solvers'
 * work-vector arena.  *bytes_released may be NULL. */
int32_t kryst_ctx_trim(kryst_ctx_t ctx, int64_t* bytes_released);
/* measurement only: per-phase device time of the work enqueued between begin and end (hipEvents recorded on the compute stream
 * after each phase: time between two marks is charged to the later one).  ms[p] for p < kryst_phase_count(): "spmv" (tiles
 * without halo columns; single rank: the whole SpMV), "halo_wait" (compute stream waiting for the neighbour planes),
 * "spmv_boundary", "reduce" (tile-partial fold + RCCL all-gather + rank-ordered fold + scalar step), "blas1" (vector updates other than
 * the next two), "pc", "blas1_residual" (CG / PCG: r -= alpha Ap with its fused inner products), "blas1_direction" (CG / PCG: x += alpha p,
 * p = z + beta p), "blas1_xbatch" (CG / PCG with the direction pass inside the SpMV: x += alpha_i p_i for a batch of iterations in one pass). */
int32_t kryst_phase_timing_begin(kryst_ctx_t ctx);
int32_t kryst_phase_timing_end(kryst_ctx_t ctx, double* ms, int32_t count);
int32_t kryst_phase_count(void);
const char* kryst_phase_name(int32_t phase);
/* wall-clock of the device work enqueued between the two marks, in ms (hipEvent on the ctx compute stream) */
int32_t kryst_ctx_timer_start(kryst_ctx_t ctx);
int32_t kryst_ctx_timer_stop(kryst_ctx_t ctx, double* ms);

/* ---- device vectors (the reference's V = Vec<f64>); in a distributed ctx n is the LOCAL length ---- */
int32_t kryst_vec_create(kryst_ctx_t ctx, int64_t n, kryst_vec_t* out);
int32_t kryst_vec_destroy(kryst_vec_t v);
int32_t kryst_vec_len(kryst_vec_t v, int64_t* n);
int32_t kryst_vec_upload(kryst_vec_t v, const double* host, int64_t n);
int32_t kryst_vec_download(kryst_vec_t v, double* host, int64_t n);
int32_t kryst_vec_fill(kryst_vec_t v, double value);
int32_t kryst_vec_copy(kryst_vec_t dst, kryst_vec_t src);
/* deterministic synthetic fill on device: v[i] = uniform[0,1) from splitmix64(seed, global index) */
int32_t kryst_vec_fill_splitmix(kryst_vec_t v, uint64_t seed, int64_t global_offset);

/* ---- CSR operator: CsrMatrix::from_csr (src/matrix/sparse.rs:28-46), usize = uint64 layout ---- */
int32_t kryst_csr_create(kryst_ctx_t ctx, int64_t nrows, int64_t ncols, const uint64_t* row_ptr,
                         const uint64_t* col_idx, const double* vals, kryst_csr_t* out);
/* same with int32 column indices / int64 row pointers (what the device keeps; avoids 2x host memory) */
int32_t kryst_csr_create_i32(kryst_ctx_t ctx, int64_t nrows, int64_t ncols, const int64_t* row_ptr,
                             const int32_t* col_idx, const double* vals, kryst_csr_t* out);
/* Row-partitioned operator: this rank owns global rows [row_lo,row_hi) (= row_offsets[rank..rank+1]);
 * row_ptr is local (row_ptr[0] == 0), col_idx are GLOBAL columns.  Builds the halo exchange plan (one
 * RCCL exchange of index lists).  The reference has no counterpart (mpi_comm.rs:133-143 is a TODO). */
int32_t kryst_csr_create_dist(kryst_ctx_t ctx, int64_t n_global, const int64_t* row_offsets /*nranks+1*/,
                              const int64_t* row_ptr, const int64_t* col_idx_global, const double* vals,
                              kryst_csr_t* out);
/* Synthetic 7-point stencil operator generated on the device: kind 0 Poisson, 1 anisotropic,
 * 2 upwind convection-diffusion, 3 symmetric variable-coefficient diffusion (per-edge weights from splitmix64: no two rows
 * alike, so no value dictionary / row patterns apply -- what a structured grid with real coefficients looks like);
 * grid N^3; in a distributed ctx each rank builds its k-slab. */
int32_t kryst_csr_create_stencil7(kryst_ctx_t ctx, int32_t N, int32_t kind, kryst_csr_t* out);
int32_t kryst_csr_destroy(kryst_csr_t a);
int32_t kryst_csr_shape(kryst_csr_t a, int64_t* nrows_local, int64_t* ncols_global, int64_t* nnz_local);
/* storage form kryst_spmv streams for this operator (all forms are lossless re-encodings made at creation beside the CSR arrays;
 * results are bit-identical): 0 plain CSR, 1 CSR-D8 (1-byte column-offset codes), 2 CSR-D16 (offset + value codes, 2 B per
 * entry), 3 CSR-P16 (one 16-bit row-pattern id per row), 4 CSR-DIA (one value stream per diagonal: operators with at most 32
 * well-filled diagonals that have no D16 / P16 form, e.g. variable-coefficient stencils).  patterns / table_entries (may be NULL):
 * size of the P16 tables. */
int32_t kryst_csr_encoding(kryst_csr_t a, int32_t* encoding, int32_t* patterns, int32_t* table_entries);
/* measurement hook: the order in which an operator's 512-row tiles are handed to the XCDs (plane-structured operators walk the
 * plane segment by segment so that an XCD's L2 keeps its window of x; which rows a tile holds and every result bit are unchanged).
 * info[0] rows per plane (0: natural order only), info[1], info[2] slots of the two orders, info[3] 1 if kryst_spmv uses it now */
int32_t kryst_csr_tile_order(kryst_csr_t a, int64_t* info);
/* measurement hook: the staged-window form of the CSR-P16 kernel (operators whose row patterns are (far, -n, -1, 0, +1, +n, far) with one
 * even n <= 1024: the near operands of a run of tiles come out of an LDS window).  info[0] n (0: not this form), info[1] 1 if the far
 * offsets are the same in every pattern, info[2] first tile of a rank's contiguous interior range (-1: none), info[3] 1 if kryst_spmv
 * takes that kernel now */
int32_t kryst_csr_pattern_info(kryst_csr_t a, int64_t* info);
int32_t kryst_csr_download(kryst_csr_t a, int64_t* row_ptr, int32_t* col_idx_local, double* vals);
/* Measurement hook (ABI 5): where the CSR arrays live.  The same plain-CSR stream mix runs at 0.70 .. 0.76 of the HBM peak depending on where
 * the driver put the three arrays (round 4, 512^3), so a single-rank creation may try several homes for (row_ptr, col, val) -- K =
 * KRYST_CSR_PLACEMENT_TRIES, default 3 once the arrays exceed 4 GB, else 1 -- time the kernel's traffic skeleton on each and keep the fastest.
 * *tries homes tried, *chosen the one kept, skeleton_ms8[8] the skeleton's milliseconds per launch on each home tried. */
int32_t kryst_csr_placement_info(kryst_csr_t a, int32_t* tries, int32_t* chosen, double* skeleton_ms8);

/* MatVec::matvec (src/core/traits.rs:4-7) == SparseMatrix::spmv (sparse.rs:56-67): y <- A x, y overwritten */
int32_t kryst_spmv(kryst_csr_t a, kryst_vec_t x, kryst_vec_t y);
/* operator-level drop-in on host slices (PCIe both ways; plumbing / Jacobi::setup-style callers only) */
int32_t kryst_spmv_host(kryst_csr_t a, const double* x, int64_t nx, double* y, int64_t ny);

/* measurement hook: `reps` back-to-back launches of the SpMV kernel (fused_dots = 0 plain, 1 = the CG kernel
 * with the (x,Ax) partials, 2 = BiCGStab's) between two HIP events on the compute stream; average ms per launch */
int32_t kryst_bench_spmv(kryst_csr_t a, kryst_vec_t x, kryst_vec_t y, int32_t fused_dots, int32_t reps, double* avg_ms);
/* measurement only: average time of one pass of a BLAS-1 stream shape (kind 0: Gram-Schmidt link, 3 vectors; 1: eight batched
 * dots, 9 vectors; 2: CG x/r update, 4 vectors; 3-5: the same with forced nontemporal loads/stores; 6: CG direction update,
 * 2 vectors; 7: CG residual pass r -= a q with (r,r), 2 vectors; 8: CG direction pass with the deferred x update, 3 vectors)
 * over vectors of n doubles placed stride_bytes apart in one allocation */
int32_t kryst_bench_streams(kryst_ctx_t ctx, int64_t n, int64_t stride_bytes, int32_t kind, int32_t reps, double* avg_ms);
/* measurement only: the plain-CSR SpMV's TRAFFIC without its arithmetic on the operator's own CSR arrays -- row pointers, values and
 * column indices streamed, x read once, y written once (SURVEY 8(d)'s bytes; y receives garbage): what this mix of five read streams and
 * one written reaches on this HBM, for `roofline_csr`'s "fraction of what the mix can reach" (bench.py: stream_skeleton) */
int32_t kryst_bench_csr_skeleton(kryst_csr_t a, kryst_vec_t x, kryst_vec_t y, int32_t reps, double* avg_ms);
/* measurement only (ABI 5): average ms per launch of the kernel CG / PCG launch when the direction pass rides inside the SpMV (cg.rs:207-209,274-276 +
 * sparse.rs:107-113 in one pass: z = x, p_old, x-update vectors of its own); KRYST_UNSUPPORTED when the operator has no staged CSR-P16 form */
int32_t kryst_bench_spmv_fused(kryst_csr_t a, kryst_vec_t x, kryst_vec_t y, int32_t reps, double* avg_ms);
/* test hook: fills the LDS of every compute unit with NaNs.  LDS is not cleared between kernels, so whatever a kernel reads from LDS
 * before writing it is what an earlier kernel -- of any process -- left there; a round-4 kernel did, and was wrong on one GPU box in five */
int32_t kryst_bench_poison_lds(kryst_ctx_t ctx);

/* ---- BLAS-1: InnerProduct for () (src/core/wrappers.rs:90-127) and the solvers' pointwise loops ---- */
int32_t kryst_dot(kryst_vec_t x, kryst_vec_t y, double* out);       /* wrappers.rs:90-108 */
int32_t kryst_norm(kryst_vec_t x, double* out);                     /* wrappers.rs:110-127 */
int32_t kryst_axpy(double alpha, kryst_vec_t x, kryst_vec_t y);     /* y[i] = y[i] + alpha*x[i]   cg.rs:207-209 */
int32_t kryst_aypx(double beta, kryst_vec_t x, kryst_vec_t y);      /* y[i] = x[i] + beta*y[i]    cg.rs:274-276 */
int32_t kryst_sub(kryst_vec_t a, kryst_vec_t b, kryst_vec_t out);   /* out[i] = a[i] - b[i]       cg.rs:123 */

/* ---- preconditioners: Preconditioner<M,V>::{setup,apply} (src/preconditioner/mod.rs:8-13) ---- */
enum { KRYST_ILU_KRYST_COMPAT = 0,   /* Ilu0 exactly as written, src/preconditioner/ilu.rs:59-122 */
       KRYST_ILU_ILUP0 = 1,          /* Ilup::new(0) exactly as written, src/preconditioner/ilup.rs:77-167 */
       KRYST_ILU_TRUE_ILU0 = 2 };    /* extension: textbook ILU(0) on A's pattern */
int32_t kryst_pc_identity(kryst_ctx_t ctx, kryst_pc_t* out);                        /* test IdentityPC, pcg.rs:245-251 */
int32_t kryst_pc_jacobi(kryst_csr_t a, kryst_pc_t* out);                            /* Jacobi::setup jacobi.rs:53-73 */
int32_t kryst_pc_ilu0(kryst_csr_t a, int32_t mode, kryst_pc_t* out);                /* Ilu0::setup / Ilup::setup */
/* Ilup::new(fill).setup(a), src/preconditioner/ilup.rs:77-134 exactly as written (level-of-fill p), on sparse rows */
int32_t kryst_pc_ilup(kryst_csr_t a, int32_t fill, kryst_pc_t* out);
/* Ilut::new(fill, droptol).setup(a), src/preconditioner/ilut.rs:80-117 exactly as written */
int32_t kryst_pc_ilut(kryst_csr_t a, int32_t fill, double droptol, kryst_pc_t* out);
int32_t kryst_pc_chebyshev_stub(kryst_ctx_t ctx, int32_t degree, kryst_pc_t* out);  /* Chebyshev trait object: apply -> SolveError, chebyshev.rs:68-70 */
int32_t kryst_pc_chebyshev(kryst_csr_t a, double alpha, double beta, int32_t degree, kryst_pc_t* out); /* extension: apply == apply_chebyshev */
/* ApproxInv with GIVEN inverse rows (ApproxInv::inv_rows, approxinv.rs:66): apply (approxinv.rs:268-298) is the sparse-row
 * product z = M r, i.e. kryst_spmv with M.  m is borrowed (must outlive the preconditioner).  ApproxInv::setup -- a
 * least-squares fit per column through faer's QR (approxinv.rs:129-264) -- stays on the host with the reference. */
int32_t kryst_pc_approx_inverse(kryst_csr_t m, kryst_pc_t* out);
int32_t kryst_pc_apply(kryst_pc_t pc, kryst_vec_t r, kryst_vec_t z);                /* Preconditioner::apply */
int32_t kryst_pc_destroy(kryst_pc_t pc);
/* measurement hooks (bench.py): average ms of `reps` back-to-back applies between two HIP events on the compute stream; and what
 * an ILU-family preconditioner's apply runs and streams -- info[0] form (0 level-ordered, 1 grid 8 x 8 lines per workgroup,
 * 2 grid 16 x 16, 3 plane kernels after a give-up), [1..3] Ni Nj Nk, [4..5] dependency levels of L / U, [6..7] coefficient chunks
 * forward / backward, [8..9] of which repeat and are not requested, [10..11] bytes per chunk request, [12] reserved; count >= 13 */
int32_t kryst_bench_pc_apply(kryst_pc_t pc, kryst_vec_t r, kryst_vec_t z, int32_t reps, double* avg_ms);
int32_t kryst_pc_ilu_info(kryst_pc_t pc, int64_t* info, int32_t count);
/* apply_chebyshev(a, r, z, alpha, beta, m), src/preconditioner/chebyshev.rs:83-140 */
int32_t kryst_apply_chebyshev(kryst_csr_t a, kryst_vec_t r, kryst_vec_t z, double alpha, double beta, int64_t m);

/* ---- solvers: LinearSolver<M,V>::solve (src/solver/mod.rs:30-52) ---- */
typedef struct {
    double  tol;                 /* Convergence::tol      src/utils/convergence.rs:4-7 */
    int64_t max_iters;           /* Convergence::max_iters */
    int32_t restart;             /* GmresSolver::restart  gmres.rs:40 */
    int32_t precond_side;        /* gmres.rs:28-32 Preconditioning: 0 None, 1 Left (default), 2 Right; 3 = textbook Left, a labelled extension (kryst_gmres_solve) */
    int32_t norm_type;           /* CgNormType cg.rs:35: 0 Preconditioned, 1 Unpreconditioned (default), 2 Natural, 3 None */
    int32_t single_reduction;    /* with_single_reduction cg.rs:69 (same fold on the device; accepted, no effect) */
    int32_t has_radius;  double radius;        /* with_radius     cg.rs:74  (CG: trust-region exit cg.rs:177-202; PCG ignores it like the reference) */
    int32_t has_obj_target; double obj_target; /* with_obj_target cg.rs:79  (CG: objective exit cg.rs:231-252) */
    int32_t check_every;         /* host polls the device convergence flag every this many iterations (0 = default);
                                    the device stops at the exact reference iteration regardless */
} kryst_params_t;

typedef struct {                 /* SolveStats src/utils/convergence.rs:10-14 */
    int64_t iterations;
    double  final_residual;
    int32_t converged;
} kryst_stats_t;

typedef void (*kryst_monitor_fn)(int64_t iteration, double residual, void* user);   /* with_monitor cg.rs:84 */

/* Residual history: hist[0..min(*hist_len,hist_cap)) receives what CgSolver/PcgSolver push to
 * residual_history (cg.rs:140,263; pcg.rs:146,199); GMRES / BiCGStab record |g[j+1]| / ||r|| per iteration
 * (an addition: the reference keeps none).  At most 2^22 entries are recorded per solve; *hist_len counts every push.
 * monitor (if non-NULL) is LIVE like the reference's (cg.rs:137-140,260-263; pcg.rs:143-146,196-199): the device
 * publishes each history entry to mapped host memory and the host fires monitor(iteration, residual, user) on the
 * calling thread, in order, every time its poll loop has waited for a batch of params->check_every iterations
 * (default 8; 1 = after every iteration; GMRES / FGMRES: once per restart cycle) and for the remaining entries when
 * the solve ends -- always before the call returns.  The callback must not call into the same context. */
#define KRYST_SOLVE_ARGS kryst_csr_t a, kryst_pc_t pc /* NULL = None */, \
        const kryst_params_t* params, kryst_stats_t* stats, \
        double* hist, int64_t hist_cap, int64_t* hist_len, kryst_monitor_fn monitor, void* user

/* b, x on the host (x in/out = initial guess / solution), exactly the reference call shape */
int32_t kryst_cg_solve      (const double* b, double* x, int64_t n, KRYST_SOLVE_ARGS);  /* CgSolver::solve       cg.rs:114-288 */
int32_t kryst_pcg_solve     (const double* b, double* x, int64_t n, KRYST_SOLVE_ARGS);  /* PcgSolver::solve      pcg.rs:114-222 */
int32_t kryst_gmres_solve   (const double* b, double* x, int64_t n, KRYST_SOLVE_ARGS);  /* GmresSolver::solve    gmres.rs:216-402 */
int32_t kryst_bicgstab_solve(const double* b, double* x, int64_t n, KRYST_SOLVE_ARGS);  /* BiCgStabSolver::solve bicgstab.rs:69-293 */
/* same with b, x resident in HBM (the performant drop-in; bench.py times these) */
int32_t kryst_cg_solve_dev      (kryst_vec_t b, kryst_vec_t x, KRYST_SOLVE_ARGS);
int32_t kryst_pcg_solve_dev     (kryst_vec_t b, kryst_vec_t x, KRYST_SOLVE_ARGS);
int32_t kryst_gmres_solve_dev   (kryst_vec_t b, kryst_vec_t x, KRYST_SOLVE_ARGS);
int32_t kryst_bicgstab_solve_dev(kryst_vec_t b, kryst_vec_t x, KRYST_SOLVE_ARGS);
/* extension: right-preconditioned BiCGStab (the reference ignores pc, bicgstab.rs:70) */
int32_t kryst_bicgstab_rpc_solve_dev(kryst_vec_t b, kryst_vec_t x, KRYST_SOLVE_ARGS);

/* CgsSolver::solve (src/solver/cgs.rs:58-135) and TfqmrSolver::solve (src/solver/tfqmr.rs:64-221).  Both ignore pc like the
 * reference (cgs.rs:59, tfqmr.rs:66); TFQMR also overwrites the initial guess with zeros (tfqmr.rs:72).  History (an
 * addition): ||r|| per iteration (CGS), the residual estimate dpest per substep, two per iteration (TFQMR). */
int32_t kryst_cgs_solve      (const double* b, double* x, int64_t n, KRYST_SOLVE_ARGS);
int32_t kryst_tfqmr_solve    (const double* b, double* x, int64_t n, KRYST_SOLVE_ARGS);
int32_t kryst_cgs_solve_dev  (kryst_vec_t b, kryst_vec_t x, KRYST_SOLVE_ARGS);
int32_t kryst_tfqmr_solve_dev(kryst_vec_t b, kryst_vec_t x, KRYST_SOLVE_ARGS);
/* FgmresSolver::solve_flex (src/solver/fgmres.rs:114-340); pc plays the FlexiblePreconditioner (preconditioner/mod.rs:16-19),
 * NULL = None.  params: tol, max_iters, restart (fgmres.rs:52-54).  orthog: OrthogMethod 0 Classical (default, :59) /
 * 1 Modified; haptol: happy-breakdown tolerance (default 1e-12, :60); preallocate: set_preallocate_vectors (:77; only
 * the cycle length `m` depends on it, :203).  History = what residual_history receives (:292). */
int32_t kryst_fgmres_solve    (const double* b, double* x, int64_t n, int32_t orthog, double haptol, int32_t preallocate, KRYST_SOLVE_ARGS);
int32_t kryst_fgmres_solve_dev(kryst_vec_t b, kryst_vec_t x, int32_t orthog, double haptol, int32_t preallocate, KRYST_SOLVE_ARGS);

/* ---- stepping session: the same solver split into begin / step / end, so that a caller (bench.py) can
 * enqueue and time exactly K iterations.  method: 0 CgSolver, 1 PcgSolver, 2 BiCgStabSolver, 3 CgsSolver, 4 TfqmrSolver.
 * step() enqueues up to k further iterations (never past max_iters) without synchronising the host. ---- */
typedef struct kryst_session_s* kryst_session_t;
int32_t kryst_session_begin(int32_t method, kryst_vec_t b, kryst_vec_t x, kryst_csr_t a, kryst_pc_t pc,
                            const kryst_params_t* params, kryst_session_t* out);
int32_t kryst_session_step(kryst_session_t s, int64_t k);
/* A session that is begun must be ended (the context stays busy until then: KRYST_ERR_BUSY for every other solve).  If an ILU
 * preconditioner's wavefront solve gave up during the session (see kryst_pc_ilu0), kryst_session_end returns KRYST_SOLVE_ERROR
 * once -- the preconditioner has switched to its plane kernels and the caller repeats the session; the one-shot kryst_*_solve
 * entry points repeat the solve themselves. */
int32_t kryst_session_end(kryst_session_t s, kryst_stats_t* stats, double* hist, int64_t hist_cap, int64_t* hist_len);

/* ---- host-only helpers (no GPU needed) ---- */
/* 7-point stencil rows of planes [k_lo,k_hi) with global columns; returns nnz; pass NULL arrays to size */
int64_t kryst_host_stencil7(int32_t N, int32_t kind, int32_t k_lo, int32_t k_hi,
                            int64_t* row_ptr, int64_t* col_idx, double* vals);
/* contiguous row blocks, boundaries aligned to `align` rows (k-slabs: align = N*N) */
int32_t kryst_host_partition_rows(int64_t n, int32_t nranks, int64_t align, int64_t* row_offsets /*nranks+1*/);
/* halo plan of one rank: which global columns it must receive from each owner.
 * recv_counts[nranks]; recv_cols[sum] ascending per owner.  Returns total count (call with NULL to size). */
int64_t kryst_host_halo_recv_plan(int32_t rank, int32_t nranks, const int64_t* row_offsets,
                                  const int64_t* row_ptr, const int64_t* col_idx_global,
                                  int64_t* recv_counts, int64_t* recv_cols);

/* ---- host-side factorisations on plain host arrays: no device, no context (ABI 5).  Exactly the code kryst_pc_ilup / kryst_pc_ilut run between
 * the download of the operator's rows and the upload of the factors (kryst_amd/csrc/host_factor.cpp), for CPU-only callers, for parity tests
 * against the oracle without a GPU and for the sanitizer tier (make -C kryst_amd/csrc san SAN=thread | address,undefined).
 * Rows (row_ptr[n+1], col[nnz] as int32, val[nnz]) of an n x n block; columns >= n (halo slots of a row-partitioned operator) are dropped.
 * Results: L's strictly-lower kept entries with their multipliers, U's strictly-upper kept entries, the kept diagonal (1.0 where none is
 * kept), each row in stored = ascending-column order (Ilut: in the order ilut.rs leaves them). */
typedef struct kryst_host_factors_s* kryst_host_factors_t;
/* Ilup::new(fill).setup (src/preconditioner/ilup.rs:77-134) as a row pipeline over `threads` host threads (<= 0: up to 16) in round-robin blocks
 * of `block` rows (<= 0: 2048); any thread count and block size gives the bits of the one-thread loop.  KRYST_SOLVE_ERROR on a zero u_jj
 * (ilup.rs:108-110), the column j of the LOWEST row that met one through kryst_hip_last_error_row(). */
int32_t kryst_host_ilup(int64_t n, const int64_t* row_ptr, const int32_t* col, const double* val, int32_t fill, int32_t threads, int64_t block,
                        kryst_host_factors_t* out);
/* Ilut::new(fill, droptol).setup (src/preconditioner/ilut.rs:80-117): drop by magnitude, keep the `fill` largest of a row, split at the diagonal */
int32_t kryst_host_ilut(int64_t n, const int64_t* row_ptr, const int32_t* col, const double* val, int32_t fill, double droptol, int32_t threads,
                        kryst_host_factors_t* out);
int32_t kryst_host_factors_sizes(kryst_host_factors_t f, int64_t* n, int64_t* nnz_l, int64_t* nnz_u);
/* any pointer may be NULL; l_ptr / u_ptr hold n + 1 entries, diag n */
int32_t kryst_host_factors_get(kryst_host_factors_t f, int64_t* l_ptr, int32_t* l_col, double* l_val, int64_t* u_ptr, int32_t* u_col, double* u_val,
                               double* diag);
int32_t kryst_host_factors_destroy(kryst_host_factors_t f);
/* The level scheduler of the general triangular solve (ilup.rs:138-167 walks rows one after the other; rows of one level are independent):
 * level[i] = 1 + the highest level among the rows that row i of a strictly-lower (forward != 0: rows ascending) or strictly-upper (rows
 * descending) factor depends on, 0 when it depends on none.  *nlevels (may be NULL): the number of levels. */
int32_t kryst_host_levels(int64_t n, const int64_t* ptr, const int32_t* col, int32_t forward, int32_t* level, int32_t* nlevels);

/* Matrix Market coordinate file -> CSR (0-based, rows sorted, symmetric / skew-symmetric storage expanded, duplicates summed;
 * real, integer and pattern fields).  Returns nnz, or -1 (kryst_hip_last_error() says why).  Call with NULL arrays to size,
 * then with row_ptr[nrows+1], col_idx[nnz], vals[nnz].  The reference has no file I/O. */
int64_t kryst_host_read_matrix_market(const char* path, int64_t* nrows, int64_t* ncols, int64_t* row_ptr,
                                      int64_t* col_idx, double* vals);
/* PETSc binary AIJ matrix (MatView with a binary viewer: big-endian header 1211216, rows, cols, nnz, row lengths, columns,
 * values) -> CSR, same calling convention. */
int64_t kryst_host_read_petsc_binary(const char* path, int64_t* nrows, int64_t* ncols, int64_t* row_ptr,
                                     int64_t* col_idx, double* vals);

#ifdef __cplusplus
}
#endif
#endif
