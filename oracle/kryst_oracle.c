/*
 * kryst_oracle.c -- CPU ORACLE (test infrastructure only; see kryst_oracle.h for the contract).
 * Every function cites the reference lines it restates (paths relative to /root/reference).
 * Build: gcc -O2 -ffp-contract=off -fopenmp (oracle/Makefile).  -ffp-contract=off is REQUIRED:
 * the reference never fuses a*b+c.
 */
#include "kryst_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>
#ifdef _OPENMP
#include <omp.h>
#endif

static int32_t g_threads = 1;
void kro_set_threads(int32_t n) { g_threads = n < 1 ? 1 : n; }
int32_t kro_get_threads(void) { return g_threads; }

static double* dalloc(int64_t n) { return (double*)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1)); }
static double* dzeros(int64_t n) { return (double*)calloc((size_t)(n > 0 ? n : 1), sizeof(double)); }

/* ------------------------------------------------------------------ inner products */

/* wrappers.rs:101-107 -- x.iter().zip(y).map(|(a,b)| a*b).fold(0, |acc,v| acc+v) */
static double dot_serial(const double* x, const double* y, int64_t n) {
    double acc = 0.0;
    for (int64_t i = 0; i < n; ++i) acc = acc + x[i] * y[i];
    return acc;
}

/* 64-lane xor butterfly: lane l computes a[l] + a[l ^ off] for off = 32,16,8,4,2,1 (the HIP
 * kernels' __shfl_xor sequence); IEEE addition commutes so every lane ends with the same bits. */
static double butterfly64(const double* in) {
    double a[64], b[64];
    memcpy(a, in, sizeof a);
    for (int off = 32; off >= 1; off >>= 1) {
        for (int l = 0; l < 64; ++l) b[l] = a[l] + a[l ^ off];
        memcpy(a, b, sizeof a);
    }
    return a[0];
}

/* threads[0..nthreads) -> butterfly per 64-lane wave, then serial fold across waves */
static double block_reduce(const double* threads, int nthreads) {
    int nw = nthreads / 64;
    double s = butterfly64(threads);
    for (int w = 1; w < nw; ++w) s = s + butterfly64(threads + 64 * w);
    return s;
}

/* one part (one rank's slice) in the tiled device order */
static double dot_tiled_part(const kro_reduce_t* rs, const double* x, const double* y, int64_t n) {
    const int T = rs->T, V = rs->V, F = rs->F;
    const int64_t tile = (int64_t)T * V;
    const int64_t ntiles = (n + tile - 1) / tile;
    double* partial = dalloc(ntiles);
#pragma omp parallel for schedule(static) num_threads(g_threads)
    for (int64_t q = 0; q < ntiles; ++q) {
        double th[1024];
        for (int t = 0; t < T; ++t) {
            double acc = 0.0;
            for (int v = 0; v < V; ++v) {
                int64_t i = q * tile + (int64_t)t * V + v;
                if (i < n) acc = acc + x[i] * y[i];
            }
            th[t] = acc;
        }
        partial[q] = block_reduce(th, T);
    }
    /* two-level fold of the tile partials (kryst_amd/csrc/common.h fold2): chunk c = partials [cF,(c+1)F) -- thread t
     * takes partial cF+t (0.0 past the end), butterfly, serial over waves; with more than one chunk the chunk values are
     * folded by F threads (thread t: chunks t, t+F, ... ascending), butterfly, serial over waves. */
    const int64_t nchunks = ntiles > F ? (ntiles + F - 1) / F : 1;
    double* chunk = dalloc(nchunks);
    double th[1024];
    for (int64_t c = 0; c < nchunks; ++c) {
        for (int t = 0; t < F; ++t) { int64_t i = c * F + t; th[t] = (i < ntiles) ? partial[i] : 0.0; }
        chunk[c] = block_reduce(th, F);
    }
    double r;
    if (nchunks == 1) r = chunk[0];
    else {
        for (int t = 0; t < F; ++t) {
            double acc = 0.0;
            for (int64_t i = t; i < nchunks; i += F) acc = acc + chunk[i];
            th[t] = acc;
        }
        r = block_reduce(th, F);
    }
    free(partial); free(chunk);
    return r;
}

double kro_dot(const kro_reduce_t* rs, const double* x, const double* y, int64_t n) {
    if (!rs || rs->mode == KRO_REDUCE_SERIAL) return dot_serial(x, y, n);
    if (rs->nparts <= 1) return dot_tiled_part(rs, x, y, n);
    double total = 0.0;
    for (int p = 0; p < rs->nparts; ++p) {
        int64_t lo = rs->part_off[p], hi = rs->part_off[p + 1];
        double r = dot_tiled_part(rs, x + lo, y + lo, hi - lo);
        total = (p == 0) ? r : total + r;
    }
    return total;
}

/* wrappers.rs:120-126 -- fold of x*x, then sqrt */
double kro_norm(const kro_reduce_t* rs, const double* x, int64_t n) { return sqrt(kro_dot(rs, x, x, n)); }

/* ------------------------------------------------------------------ SpMV */

/* sparse.rs:103-114 spmv_parallel: per row `sum = 0; for j in 0..ncols { sum = sum + dense[i,j]*x[j] }`.
 * The densified zeros contribute sum + 0*x[j] == sum, so the CSR row loop over ascending stored columns is
 * bit-identical for finite x (also the order of wrappers.rs:31-36, which every solver test exercises). */
void kro_spmv(const kro_csr_t* a, const double* x, double* y) {
    const int64_t n = a->nrows;
#pragma omp parallel for schedule(static) num_threads(g_threads)
    for (int64_t i = 0; i < n; ++i) {
        double sum = 0.0;
        for (int64_t k = a->row_ptr[i]; k < a->row_ptr[i + 1]; ++k)
            sum = sum + a->vals[k] * x[a->col_idx[k]];
        y[i] = sum;
    }
}

/* sparse.rs:36-42: SymbolicSparseRowMat::new_checked preconditions (in-bounds, sorted, unique) */
int32_t kro_csr_check(const kro_csr_t* a) {
    if (a->nrows < 0 || a->ncols < 0 || a->row_ptr[0] != 0) return 1;
    for (int64_t i = 0; i < a->nrows; ++i) {
        if (a->row_ptr[i + 1] < a->row_ptr[i]) return 2;
        for (int64_t k = a->row_ptr[i]; k < a->row_ptr[i + 1]; ++k) {
            if (a->col_idx[k] < 0 || a->col_idx[k] >= a->ncols) return 3;
            if (k > a->row_ptr[i] && a->col_idx[k] <= a->col_idx[k - 1]) return 4;
        }
    }
    return 0;
}

/* pointwise helpers; every expression is written exactly as in the cited reference line */
#define PFOR(i, n) _Pragma("omp parallel for schedule(static) num_threads(g_threads)") for (int64_t i = 0; i < (n); ++i)

/* r = b - A x   (cg.rs:120-125: `bi - ax`) */
static void residual(const kro_csr_t* a, const double* b, const double* x, double* r, double* tmp) {
    kro_spmv(a, x, tmp);
    PFOR(i, a->nrows) r[i] = b[i] - tmp[i];
}

/* ------------------------------------------------------------------ convergence.rs:18-34 */
static int conv_check(double tol, int64_t max_iters, double res, double res0, int64_t i, kro_stats_t* s) {
    double rel = res / res0;
    int converged = (rel <= tol) || (i >= max_iters);
    s->iterations = i; s->final_residual = res; s->converged = converged;
    return converged;
}

static void trace_push(kro_trace_t* tr, int64_t it, double v) {
    if (!tr) return;
    if (tr->monitor) tr->monitor(it, v, tr->user);
    if (tr->hist && tr->len < tr->cap) tr->hist[tr->len] = v;
    tr->len++;
}

/* ------------------------------------------------------------------ preconditioners */

static int64_t find_diag(const kro_csr_t* a, int64_t i) {
    for (int64_t k = a->row_ptr[i]; k < a->row_ptr[i + 1]; ++k) if (a->col_idx[k] == i) return k;
    return -1;
}

/* jacobi.rs:53-73: diag[i] = (A e_i)[i] = 0 + a_ii*1 (+ zeros) = a_ii exactly; inv = d != 0 ? 1/d : 0 */
int32_t kro_jacobi_setup(const kro_csr_t* a, double* inv_diag) {
    for (int64_t i = 0; i < a->nrows; ++i) {
        int64_t k = find_diag(a, i);
        double d = (k >= 0) ? (0.0 + a->vals[k] * 1.0) : 0.0;
        inv_diag[i] = (d != 0.0) ? 1.0 / d : 0.0;
    }
    return KRO_OK;
}

/* ilu.rs:59-100 as written.  Step i re-seeds row i of U from a (ilu.rs:66-72) and column i of L from
 * a[j][i]/u[i][i] (ilu.rs:76-80); the Schur update (ilu.rs:82-95) always starts from the ORIGINAL a and is
 * overwritten by those re-seeds, so the net result is  U = triu(A),  L = I + tril(A,-1) D^-1  with
 * l[j][i] = a[j][i] / a[i][i].  Stored zeros are skipped exactly as `!= T::zero()` does. */
int32_t kro_ilu0_compat_setup(const kro_csr_t* a, double* lfac, double* ufac) {
    for (int64_t i = 0; i < a->nrows; ++i)
        for (int64_t k = a->row_ptr[i]; k < a->row_ptr[i + 1]; ++k) {
            int64_t j = a->col_idx[k];
            lfac[k] = 0.0; ufac[k] = 0.0;
            if (j < i) {
                if (a->vals[k] != 0.0) {
                    int64_t kd = find_diag(a, j);
                    double ujj = (kd >= 0) ? a->vals[kd] : 0.0;
                    lfac[k] = a->vals[k] / ujj;
                }
            } else {
                ufac[k] = a->vals[k];
            }
        }
    return KRO_OK;
}

/* ilup.rs:77-134 with fill = 0 as written: every update has new_level = 0+0+1 > fill (ilup.rs:115-116), so no
 * elimination happens; l_ij = a_ij / a_jj (ilup.rs:104-111, Err on zero u_jj), U = nonzeros of row i, k >= i. */
int32_t kro_ilup0_setup(const kro_csr_t* a, double* lfac, double* ufac) {
    for (int64_t i = 0; i < a->nrows; ++i)
        for (int64_t k = a->row_ptr[i]; k < a->row_ptr[i + 1]; ++k) {
            int64_t j = a->col_idx[k];
            lfac[k] = 0.0; ufac[k] = 0.0;
            if (j < i) {
                if (a->vals[k] != 0.0) {
                    int64_t kd = find_diag(a, j);
                    double ujj = (kd >= 0) ? a->vals[kd] : 0.0;
                    if (ujj == 0.0) return KRO_SOLVE_ERROR;      /* ilup.rs:106-108 */
                    lfac[k] = a->vals[k] / ujj;
                }
            } else {
                ufac[k] = a->vals[k];
            }
        }
    return KRO_OK;
}

/* EXTENSION (not in the reference): textbook ILU(0), IKJ variant restricted to A's pattern (Saad Alg. 10.4).
 * for i: for k<i in pattern (ascending): a_ik /= a_kk; for j>k in pattern of row i: a_ij -= a_ik*a_kj. */
int32_t kro_ilu0_true_setup(const kro_csr_t* a, double* lfac, double* ufac) {
    const int64_t n = a->nrows, nnz = a->row_ptr[n];
    double* w = dalloc(nnz);
    memcpy(w, a->vals, sizeof(double) * (size_t)nnz);
    int64_t* pos = (int64_t*)malloc(sizeof(int64_t) * (size_t)(a->ncols > 0 ? a->ncols : 1));
    for (int64_t c = 0; c < a->ncols; ++c) pos[c] = -1;
    int32_t rc = KRO_OK;
    for (int64_t i = 0; i < n && rc == KRO_OK; ++i) {
        for (int64_t k = a->row_ptr[i]; k < a->row_ptr[i + 1]; ++k) pos[a->col_idx[k]] = k;
        for (int64_t k = a->row_ptr[i]; k < a->row_ptr[i + 1]; ++k) {
            int64_t c = a->col_idx[k];
            if (c >= i) break;
            int64_t kd = find_diag(a, c);
            if (kd < 0 || w[kd] == 0.0) { rc = KRO_ZERO_PIVOT; break; }
            w[k] = w[k] / w[kd];
            for (int64_t kk = a->row_ptr[c]; kk < a->row_ptr[c + 1]; ++kk) {
                int64_t j = a->col_idx[kk];
                if (j > c && pos[j] >= 0) w[pos[j]] = w[pos[j]] - w[k] * w[kk];
            }
        }
        for (int64_t k = a->row_ptr[i]; k < a->row_ptr[i + 1]; ++k) pos[a->col_idx[k]] = -1;
    }
    for (int64_t i = 0; i < n; ++i)
        for (int64_t k = a->row_ptr[i]; k < a->row_ptr[i + 1]; ++k) {
            int64_t j = a->col_idx[k];
            lfac[k] = (j < i) ? w[k] : 0.0;
            ufac[k] = (j >= i) ? w[k] : 0.0;
        }
    free(w); free(pos);
    return rc;
}

/* Triangular applies.
 *  ilu.rs:105-122 (divide_diag = 0): y1 = x; fwd: y1[i] -= l[i][j]*y1[j], j ascending < i;
 *                                    bwd: i descending, y1[i] -= u[i][j]*y1[j], j ascending > i; NO diagonal divide.
 *  ilup.rs:138-167 (divide_diag = 1): fwd: sum = r[i]; sum -= l_ij*y[j] in stored order; bwd: sum = y[i];
 *                                    sum -= u_ij*z[j] (j > i, stored order); z[i] = sum / u_ii when stored.
 * Skipped zero entries contribute t - 0*y == t, so the sparse loop equals the dense one for finite data. */
static void tri_apply(const kro_pc_t* pc, const double* r, double* z, int64_t n) {
    const kro_csr_t* a = pc->a;
    for (int64_t i = 0; i < n; ++i) {
        double s = r[i];
        for (int64_t k = a->row_ptr[i]; k < a->row_ptr[i + 1]; ++k) {
            int64_t j = a->col_idx[k];
            if (j >= i) break;
            if (pc->lfac[k] != 0.0) s = s - pc->lfac[k] * z[j];
        }
        z[i] = s;
    }
    for (int64_t i = n - 1; i >= 0; --i) {
        double s = z[i];
        double d = 0.0; int has_d = 0;
        for (int64_t k = a->row_ptr[i]; k < a->row_ptr[i + 1]; ++k) {
            int64_t j = a->col_idx[k];
            if (j > i) { if (pc->ufac[k] != 0.0) s = s - pc->ufac[k] * z[j]; }
            else if (j == i && pc->ufac[k] != 0.0) { d = pc->ufac[k]; has_d = 1; }
        }
        z[i] = (pc->divide_diag && has_d) ? s / d : s;
    }
}

/* ---- Ilup(p) and Ilut as written ------------------------------------------------------------------------ */
typedef struct { int64_t* col; double* val; int64_t len, cap; } rowbuf_t;
static void rb_push(rowbuf_t* r, int64_t c, double v) {
    if (r->len == r->cap) { r->cap = r->cap ? 2 * r->cap : 8; r->col = realloc(r->col, sizeof(int64_t) * r->cap); r->val = realloc(r->val, sizeof(double) * r->cap); }
    r->col[r->len] = c; r->val[r->len] = v; r->len++;
}
static void rows_to_tri(int64_t n, rowbuf_t* l, rowbuf_t* u, kro_trirows_t* out) {
    out->n = n;
    out->l_ptr = malloc(sizeof(int64_t) * (n + 1)); out->u_ptr = malloc(sizeof(int64_t) * (n + 1));
    int64_t nl = 0, nu = 0;
    for (int64_t i = 0; i < n; ++i) { nl += l[i].len; nu += u[i].len; }
    out->l_col = malloc(sizeof(int64_t) * (nl + 1)); out->l_val = malloc(sizeof(double) * (nl + 1));
    out->u_col = malloc(sizeof(int64_t) * (nu + 1)); out->u_val = malloc(sizeof(double) * (nu + 1));
    nl = nu = 0; out->l_ptr[0] = out->u_ptr[0] = 0;
    for (int64_t i = 0; i < n; ++i) {
        for (int64_t k = 0; k < l[i].len; ++k) { out->l_col[nl] = l[i].col[k]; out->l_val[nl++] = l[i].val[k]; }
        for (int64_t k = 0; k < u[i].len; ++k) { out->u_col[nu] = u[i].col[k]; out->u_val[nu++] = u[i].val[k]; }
        out->l_ptr[i + 1] = nl; out->u_ptr[i + 1] = nu;
        free(l[i].col); free(l[i].val); free(u[i].col); free(u[i].val);
    }
}
void kro_trirows_free(kro_trirows_t* t) {
    free(t->l_ptr); free(t->l_col); free(t->l_val); free(t->u_ptr); free(t->u_col); free(t->u_val);
    memset(t, 0, sizeof *t);
}

/* ilup.rs:77-134, dense level[][] and a_work[][] as written */
int32_t kro_ilup_build(const kro_csr_t* a, int64_t fill, kro_trirows_t* out) {
    const int64_t n = a->nrows;
    const uint64_t UMAX = UINT64_MAX;
    uint64_t* level = malloc(sizeof(uint64_t) * (size_t)(n * n + 1));
    double* w = calloc((size_t)(n * n + 1), sizeof(double));
    for (int64_t k = 0; k < n * n; ++k) level[k] = UMAX;
    for (int64_t i = 0; i < n; ++i)
        for (int64_t k = a->row_ptr[i]; k < a->row_ptr[i + 1]; ++k) {
            w[i * n + a->col_idx[k]] = a->vals[k];
            if (a->vals[k] != 0.0) level[i * n + a->col_idx[k]] = 0;                       /* :88-94 */
        }
    rowbuf_t* l = calloc((size_t)n + 1, sizeof(rowbuf_t)); rowbuf_t* u = calloc((size_t)n + 1, sizeof(rowbuf_t));
    int32_t rc = KRO_OK;
    for (int64_t i = 0; i < n && rc == KRO_OK; ++i) {                                       /* :103 */
        for (int64_t j = 0; j < i; ++j) {
            if (w[i * n + j] != 0.0 && level[i * n + j] <= (uint64_t)fill) {                /* :106 */
                const double u_jj = w[j * n + j];
                if (u_jj == 0.0) { rc = KRO_SOLVE_ERROR; break; }                           /* :108-110 */
                const double lij = w[i * n + j] / u_jj;
                rb_push(&l[i], j, lij);
                for (int64_t k = j + 1; k < n; ++k)
                    if (w[j * n + k] != 0.0) {                                              /* :117 */
                        uint64_t nl = level[i * n + j];                                     /* saturating_add chain :118 */
                        nl = (nl > UMAX - level[j * n + k]) ? UMAX : nl + level[j * n + k];
                        nl = (nl == UMAX) ? UMAX : nl + 1;
                        if (nl <= (uint64_t)fill) {
                            const double update = lij * w[j * n + k];
                            w[i * n + k] = w[i * n + k] - update;
                            if (nl < level[i * n + k]) level[i * n + k] = nl;
                        }
                    }
            }
        }
        for (int64_t k = i; k < n; ++k)
            if (w[i * n + k] != 0.0 && level[i * n + k] <= (uint64_t)fill) rb_push(&u[i], k, w[i * n + k]);   /* :129-134 */
    }
    rows_to_tri(n, l, u, out);
    free(l); free(u); free(level); free(w);
    if (rc != KRO_OK) kro_trirows_free(out);
    return rc;
}

/* ilut.rs:80-117: no elimination -- drop by magnitude, keep the `fill` largest (stable descending sort), split at the diagonal */
int32_t kro_ilut_build(const kro_csr_t* a, int64_t fill, double droptol, kro_trirows_t* out) {
    const int64_t n = a->nrows;
    rowbuf_t* l = calloc((size_t)n + 1, sizeof(rowbuf_t)); rowbuf_t* u = calloc((size_t)n + 1, sizeof(rowbuf_t));
    for (int64_t i = 0; i < n; ++i) {
        rowbuf_t row = {0};
        for (int64_t k = a->row_ptr[i]; k < a->row_ptr[i + 1]; ++k)
            if (a->vals[k] != 0.0 && fabs(a->vals[k]) >= droptol) rb_push(&row, a->col_idx[k], a->vals[k]);   /* :88-95 */
        if (row.len > fill) {                                                               /* :97-100 stable sort by |v| descending */
            for (int64_t p = 1; p < row.len; ++p) {                                         /* insertion sort == stable */
                const int64_t c = row.col[p]; const double v = row.val[p];
                int64_t q = p - 1;
                while (q >= 0 && fabs(row.val[q]) < fabs(v)) { row.col[q + 1] = row.col[q]; row.val[q + 1] = row.val[q]; --q; }
                row.col[q + 1] = c; row.val[q + 1] = v;
            }
            row.len = fill;
        }
        for (int64_t k = 0; k < row.len; ++k) {                                             /* :104-112 */
            if (row.col[k] < i) rb_push(&l[i], row.col[k], row.val[k]);
            else rb_push(&u[i], row.col[k], row.val[k]);
        }
        free(row.col); free(row.val);
    }
    rows_to_tri(n, l, u, out);
    free(l); free(u);
    return KRO_OK;
}

/* ilup.rs:138-167 == ilut.rs:121-150 */
static void trirows_apply(const kro_trirows_t* t, const double* r, double* z) {
    const int64_t n = t->n;
    double* y = dzeros(n);
    for (int64_t i = 0; i < n; ++i) {
        double sum = r[i];
        for (int64_t k = t->l_ptr[i]; k < t->l_ptr[i + 1]; ++k) sum = sum - t->l_val[k] * y[t->l_col[k]];
        y[i] = sum;
    }
    for (int64_t i = n - 1; i >= 0; --i) {
        double sum = y[i];
        int64_t dpos = -1;
        for (int64_t k = t->u_ptr[i]; k < t->u_ptr[i + 1]; ++k) {
            if (t->u_col[k] > i) sum = sum - t->u_val[k] * z[t->u_col[k]];
            else if (t->u_col[k] == i && dpos < 0) dpos = k;                                /* .position(|col| col == i): first match */
        }
        z[i] = (dpos >= 0) ? sum / t->u_val[dpos] : sum;
    }
    free(y);
}

/* chebyshev.rs:143-159 */
double kro_chebyshev_t(int64_t m, double x) {
    if (m == 0) return 1.0;
    if (m == 1) return x;
    double t0 = 1.0, t1 = x, t2;
    for (int64_t k = 2; k <= m; ++k) { t2 = 2.0 * x * t1 - t0; t0 = t1; t1 = t2; }
    return t1;
}

/* chebyshev.rs:83-140 */
void kro_apply_chebyshev(const kro_csr_t* a, const double* r, double* z, int64_t n,
                         double alpha, double beta, int64_t m) {
    if (fabs(beta - alpha) < DBL_EPSILON) { memcpy(z, r, sizeof(double) * (size_t)n); return; }  /* :88-92 */
    double* v0 = dalloc(n); double* v1 = dzeros(n); double* v2 = dzeros(n);
    memcpy(v0, r, sizeof(double) * (size_t)n);
    double c = (beta + alpha) / 2.0;
    double d = (beta - alpha) / 2.0;
    double tau = 1.0 / kro_chebyshev_t(m, (0.0 - c) / d);                                           /* :102 */
    kro_spmv(a, v0, v1);
    PFOR(i, n) v1[i] = (v1[i] - c * v0[i]) / d;                                                     /* :105-107 */
    if (m == 0) { memcpy(z, v0, sizeof(double) * (size_t)n); goto done; }                           /* :108-111 */
    if (m == 1) { memcpy(z, v1, sizeof(double) * (size_t)n); goto done; }                           /* :112-116 (unscaled) */
    for (int64_t k = 2; k <= m; ++k) {
        kro_spmv(a, v1, v2);
        PFOR(i, n) v2[i] = (2.0 * (v2[i] - c * v1[i]) / d) - v0[i];                                 /* :121 */
        double* t = v0; v0 = v1; v1 = t;       /* swap(v0,v1) */
        t = v1; v1 = v2; v2 = t;               /* swap(v1,v2) */
    }
    PFOR(i, n) z[i] = tau * v1[i];                                                                  /* :130-138 */
done:
    free(v0); free(v1); free(v2);
}

int32_t kro_pc_apply(const kro_pc_t* pc, const double* r, double* z, int64_t n) {
    switch (pc ? pc->kind : KRO_PC_NONE) {
    case KRO_PC_NONE:
    case KRO_PC_IDENTITY: memcpy(z, r, sizeof(double) * (size_t)n); return KRO_OK;
    case KRO_PC_JACOBI: { PFOR(i, n) z[i] = pc->inv_diag[i] * r[i]; return KRO_OK; }               /* jacobi.rs:84-92 */
    case KRO_PC_ILU0_COMPAT: case KRO_PC_ILUP0: case KRO_PC_ILU0_TRUE: tri_apply(pc, r, z, n); return KRO_OK;
    case KRO_PC_TRIROWS: trirows_apply(pc->rows, r, z); return KRO_OK;
    case KRO_PC_SPAI: kro_spmv(pc->a, r, z); return KRO_OK;                                         /* approxinv.rs:268-298 */
    case KRO_PC_CHEBYSHEV_STUB: return KRO_SOLVE_ERROR;                                             /* chebyshev.rs:68-70 */
    case KRO_PC_CHEBYSHEV: kro_apply_chebyshev(pc->a, r, z, n, pc->cheb_alpha, pc->cheb_beta, pc->cheb_degree); return KRO_OK;
    default: return KRO_UNSUPPORTED;
    }
}

/* ------------------------------------------------------------------ CG  (cg.rs:114-288) */
int32_t kro_cg(const kro_csr_t* a, const kro_pc_t* pc, const double* b, double* x,
               const kro_params_t* p, const kro_reduce_t* rs, kro_stats_t* st, kro_trace_t* tr) {
    (void)pc;                                                       /* cg.rs:115 `let _ = pc;` */
    const int64_t n = a->nrows;
    double* xv = dalloc(n); memcpy(xv, x, sizeof(double) * (size_t)n);
    double* r = dalloc(n); double* pp = dalloc(n); double* ap = dalloc(n); double* tmp = dalloc(n);
    int32_t rc = KRO_OK;
    residual(a, b, xv, r, tmp);                                     /* :120-125 */
    memcpy(pp, r, sizeof(double) * (size_t)n);                      /* :126 */
    double rsq = kro_dot(rs, r, r, n);                              /* :127 */
    double res0 = sqrt(rsq);                                        /* :128 */
    st->iterations = 0; st->final_residual = res0; st->converged = 0;
    double dp;                                                      /* :131-136 */
    switch (p->norm_type) {
        case 0: case 1: dp = kro_dot(rs, r, r, n); break;
        case 2: dp = kro_dot(rs, r, pp, n); break;
        default: dp = 0.0;
    }
    trace_push(tr, 0, sqrt(dp));                                    /* :137-140 */
    for (int64_t i = 1; i <= p->max_iters; ++i) {                   /* :141 */
        kro_spmv(a, pp, ap);                                        /* :143-144 */
        double p_dot_ap = kro_dot(rs, pp, ap, n);                   /* :146-165 (both variants are the same fold) */
        double res_norm;
        if (p_dot_ap <= 0.0) {                                      /* :168-174 */
            res_norm = sqrt(kro_dot(rs, r, r, n));
            st->iterations = i; st->final_residual = res_norm; st->converged = 0;
            rc = KRO_INDEFINITE_MATRIX; goto out_noupdate;          /* x is NOT written back on Err */
        }
        double alpha = rsq / p_dot_ap;                              /* :175 */
        if (p->has_radius) {                                        /* :177-202 */
            double p_norm = sqrt(kro_dot(rs, pp, pp, n));
            double x_norm = sqrt(kro_dot(rs, xv, xv, n));
            if (x_norm + fabs(alpha) * p_norm > p->radius) {
                double max_step = (p->radius - x_norm) / p_norm;
                PFOR(j, n) xv[j] = xv[j] + max_step * pp[j];
                double res_tr = sqrt(kro_dot(rs, r, r, n));
                st->iterations = i; st->final_residual = res_tr; st->converged = 0;
                goto out;
            }
        }
        PFOR(j, n) xv[j] = xv[j] + alpha * pp[j];                   /* :207-209 */
        PFOR(j, n) r[j] = r[j] - alpha * ap[j];                     /* :210-212 */
        double rsq_new = kro_dot(rs, r, r, n);                      /* :223 */
        switch (p->norm_type) {                                     /* :224-229 */
            case 0: case 1: res_norm = sqrt(rsq_new); break;
            case 2: res_norm = sqrt(fabs(kro_dot(rs, r, pp, n))); break;
            default: res_norm = 0.0;
        }
        if (p->has_obj_target) {                                    /* :231-252 */
            kro_spmv(a, xv, tmp);
            double x_dot_ax = kro_dot(rs, xv, tmp, n);
            double x_dot_b = kro_dot(rs, xv, b, n);
            double obj = 0.5 * x_dot_ax - x_dot_b;
            double res_obj;
            switch (p->norm_type) {
                case 0: res_obj = sqrt(kro_dot(rs, r, r, n)); break;
                case 1: res_obj = sqrt(rsq_new); break;
                case 2: res_obj = sqrt(fabs(kro_dot(rs, r, pp, n))); break;
                default: res_obj = 0.0;
            }
            if (obj <= p->obj_target) {
                st->iterations = i; st->final_residual = res_obj; st->converged = 1;
                goto out;
            }
        }
        if (rsq_new / rsq < 0.0) {                                  /* :254-259 */
            st->iterations = i; st->final_residual = res_norm; st->converged = 0;
            rc = KRO_INDEFINITE_PC; goto out_noupdate;
        }
        trace_push(tr, i, res_norm);                                /* :260-263 */
        int stop = conv_check(p->tol, p->max_iters, res_norm, res0, i, st);   /* :264-265 */
        if (stop && st->converged) goto out;                        /* :266-269 */
        double beta = rsq_new / rsq;                                /* :270 */
        PFOR(j, n) pp[j] = r[j] + beta * pp[j];                     /* :274-276 */
        rsq = rsq_new;                                              /* :284 */
    }
out:
    memcpy(x, xv, sizeof(double) * (size_t)n);                      /* :286 / :267 / :195 / :246 */
out_noupdate:
    free(xv); free(r); free(pp); free(ap); free(tmp);
    return rc;
}

/* ------------------------------------------------------------------ PCG  (pcg.rs:114-222) */
static double pcg_norm(const kro_reduce_t* rs, int nt, const double* r, const double* z, int64_t n, int use_abs) {
    switch (nt) {                                                   /* :137-142, :190-195 */
        case 0: return sqrt(kro_dot(rs, z, z, n));
        case 1: return sqrt(kro_dot(rs, r, r, n));
        case 2: { double d = kro_dot(rs, r, z, n); return sqrt(use_abs ? fabs(d) : d); }
        default: return 0.0;
    }
}

int32_t kro_pcg(const kro_csr_t* a, const kro_pc_t* pc, const double* b, double* x,
                const kro_params_t* p, const kro_reduce_t* rs, kro_stats_t* st, kro_trace_t* tr) {
    const int64_t n = a->nrows;
    const int has_pc = pc && pc->kind != KRO_PC_NONE;
    double* xv = dalloc(n); memcpy(xv, x, sizeof(double) * (size_t)n);
    double* r = dalloc(n); double* z = dzeros(n); double* pp = dalloc(n); double* ap = dalloc(n); double* tmp = dalloc(n);
    int32_t rc = KRO_OK;
    residual(a, b, xv, r, tmp);                                     /* :119-124 */
    if (has_pc) { rc = kro_pc_apply(pc, r, z, n); if (rc) goto out_noupdate; }   /* :127-128 `?` */
    else memcpy(z, r, sizeof(double) * (size_t)n);                  /* :130 */
    memcpy(pp, z, sizeof(double) * (size_t)n);                      /* :132 */
    double rz = kro_dot(rs, r, z, n);                               /* :133 */
    double res0 = sqrt(fabs(rz));                                   /* :134 */
    st->iterations = 0; st->final_residual = res0; st->converged = 0;
    trace_push(tr, 0, pcg_norm(rs, p->norm_type, r, z, n, 0));      /* :137-146 (no abs at iteration 0) */
    for (int64_t i = 0; i < p->max_iters; ++i) {                    /* :147 */
        kro_spmv(a, pp, ap);                                        /* :149-150 */
        double p_dot_ap = kro_dot(rs, pp, ap, n);                   /* :151-160 */
        if (p_dot_ap <= 0.0) {                                      /* :162-172 */
            st->iterations = i + 1;
            st->final_residual = pcg_norm(rs, p->norm_type, r, z, n, 1);
            st->converged = 0;
            rc = KRO_INDEFINITE_MATRIX; goto out_noupdate;
        }
        double alpha = rz / p_dot_ap;                               /* :173 */
        PFOR(j, n) xv[j] = xv[j] + alpha * pp[j];                   /* :175-177 */
        PFOR(j, n) r[j] = r[j] - alpha * ap[j];                     /* :179-181 */
        if (has_pc) { rc = kro_pc_apply(pc, r, z, n); if (rc) goto out_noupdate; }  /* :183-184 */
        else memcpy(z, r, sizeof(double) * (size_t)n);              /* :186 */
        double rz_new = kro_dot(rs, r, z, n);                       /* :188 */
        double res_norm = pcg_norm(rs, p->norm_type, r, z, n, 1);   /* :190-195 */
        trace_push(tr, i + 1, res_norm);                            /* :196-199 */
        int stop = conv_check(p->tol, p->max_iters, res_norm, res0, i + 1, st);   /* :200-201 */
        if (stop && st->converged) goto out;                        /* :202-205 */
        double beta = rz_new / rz;                                  /* :206 */
        if (beta < 0.0) {                                           /* :208-213 */
            st->iterations = i + 1; st->final_residual = res_norm; st->converged = 0;
            rc = KRO_INDEFINITE_PC; goto out_noupdate;
        }
        PFOR(j, n) pp[j] = z[j] + beta * pp[j];                     /* :215-217 */
        rz = rz_new;                                                /* :218 */
    }
out:
    memcpy(x, xv, sizeof(double) * (size_t)n);
out_noupdate:
    free(xv); free(r); free(z); free(pp); free(ap); free(tmp);
    return rc;
}

/* ------------------------------------------------------------------ GMRES (gmres.rs:216-402) */

/* gmres.rs:154-176; h is (restart+1) x restart row-major */
static void givens_update(double* h, int64_t ld, double* g, double* cs, double* sn, int64_t j, double eps) {
#define H(i, k) h[(i) * ld + (k)]
    for (int64_t i = 0; i < j; ++i) {
        double temp = cs[i] * H(i, j) + sn[i] * H(i + 1, j);
        H(i + 1, j) = -sn[i] * H(i, j) + cs[i] * H(i + 1, j);
        H(i, j) = temp;
    }
    double h_kk = H(j, j), h_k1k = H(j + 1, j);
    double r = sqrt(h_kk * h_kk + h_k1k * h_k1k);
    if (fabs(r) < eps) { cs[j] = 1.0; sn[j] = 0.0; }
    else { cs[j] = h_kk / r; sn[j] = h_k1k / r; }
    H(j, j) = cs[j] * h_kk + sn[j] * h_k1k;
    H(j + 1, j) = 0.0;
    double temp = cs[j] * g[j] + sn[j] * g[j + 1];
    g[j + 1] = -sn[j] * g[j] + cs[j] * g[j + 1];
    g[j] = temp;
}

/* gmres.rs:180-192 */
static void back_substitution(const double* h, int64_t ld, const double* g, double* y, int64_t m, double eps) {
    for (int64_t i = m - 1; i >= 0; --i) {
        y[i] = g[i];
        for (int64_t j = i + 1; j < m; ++j) y[i] = y[i] - H(i, j) * y[j];
        if (fabs(H(i, i)) > eps) y[i] = y[i] / H(i, i);
        else y[i] = 0.0;
    }
}

/* double modified Gram-Schmidt of w against basis[0..=j] (gmres.rs:83-96 / 286-298 / 318-330) */
static void mgs2(const kro_reduce_t* rs, double* w, double** basis, int64_t j, double* h, int64_t ld, int64_t n) {
    for (int64_t i = 0; i <= j; ++i) {
        double hij = kro_dot(rs, w, basis[i], n);
        H(i, j) = hij;
        const double* vi = basis[i];
        PFOR(k, n) w[k] = w[k] - hij * vi[k];
    }
    for (int64_t i = 0; i <= j; ++i) {
        double tmp = kro_dot(rs, w, basis[i], n);
        H(i, j) = H(i, j) + tmp;
        const double* vi = basis[i];
        PFOR(k, n) w[k] = w[k] - tmp * vi[k];
    }
}

int32_t kro_gmres(const kro_csr_t* a, const kro_pc_t* pc, const double* b, double* x,
                  const kro_params_t* p, const kro_reduce_t* rs, kro_stats_t* st, kro_trace_t* tr) {
    const int64_t n = a->nrows, restart = p->restart, ld = restart;
    const int has_pc = pc && pc->kind != KRO_PC_NONE;
    const int side = has_pc ? p->precond_side : 0;    /* match (self.preconditioning, pc): anything else => `_` arm */
    double* xk = dalloc(n); memcpy(xk, x, sizeof(double) * (size_t)n);
    double* r0 = dalloc(n); double* tmp = dalloc(n); double* w = dalloc(n); double* z = dalloc(n);
    double** V = (double**)calloc((size_t)restart + 1, sizeof(double*));
    double** Z = (double**)calloc((size_t)restart + 1, sizeof(double*));
    for (int64_t k = 0; k <= restart; ++k) { V[k] = dalloc(n); Z[k] = dalloc(n); }
    double* h = dalloc((restart + 1) * restart); double* g = dalloc(restart + 1);
    double* cs = dalloc(restart); double* sn = dalloc(restart); double* y = dalloc(restart);
    int32_t rc = KRO_OK;

    residual(a, b, xk, r0, tmp);                                    /* :221-226 */
    double beta = kro_norm(rs, r0, n);                              /* :227 */
    double res0 = beta;                                             /* :228 */
    st->iterations = 0; st->final_residual = beta; st->converged = 0;
    int64_t n_outer = restart > 0 ? (p->max_iters + restart - 1) / restart : 0;   /* :231 div_ceil */
    int64_t iteration = 0;
    const double eps = 1e-14;                                       /* :233 */
    double res0_in = res0;                                          /* the norm the in-cycle test divides by (side 3: ||M^-1 r0|| of the first cycle) */
    for (int64_t outer = 0; outer < n_outer; ++outer) {             /* :234 */
        int64_t nv = 0, nz = 0;
        double r0_norm = beta;                                      /* :238 */
        if (side == 3) {
            /* LABELLED EXTENSION, not in the reference: textbook left preconditioning (Saad, Alg. 9.4) in the reference's frame.
             * The reference's Left arm (:240-247, :279-307) orthogonalises against Z with the UN-normalised Z[0] = M^-1 v0 and starts g
             * from the unpreconditioned ||r0||, so its least-squares problem is not the one of M^-1 A x = M^-1 b.  Here: Arnoldi on
             * M^-1 A started from M^-1 r0 / ||M^-1 r0||, modified Gram-Schmidt twice against V as in arnoldi (:83-96), x updated with V
             * (:362-386), the in-cycle test on the preconditioned residual |g[j+1]| relative to the first cycle's ||M^-1 r0||, the
             * cycle-end test on the true residual exactly as :388-398.  Happy breakdown as in arnoldi (:97-101): Givens still applied. */
            rc = kro_pc_apply(pc, r0, z, n); if (rc) goto out_noupdate;
            r0_norm = kro_norm(rs, z, n);
            PFOR(k, n) V[0][k] = z[k] / r0_norm;
            nv = 1;
            if (outer == 0) res0_in = r0_norm;
        } else if (side == 1) {                                     /* :240-247 */
            PFOR(k, n) V[0][k] = r0[k] / r0_norm;
            nv = 1;
            rc = kro_pc_apply(pc, V[0], Z[0], n); if (rc) goto out_noupdate;   /* `.expect` panics in the reference */
            nz = 1;
        } else if (side == 2) {                                     /* :248-260 */
            rc = kro_pc_apply(pc, r0, z, n); if (rc) goto out_noupdate;
            r0_norm = kro_norm(rs, z, n);
            PFOR(k, n) V[0][k] = z[k] / r0_norm;
            nv = 1;
            rc = kro_pc_apply(pc, V[0], Z[0], n); if (rc) goto out_noupdate;
            nz = 1;
            beta = r0_norm;
        } else {                                                    /* :261-265 */
            PFOR(k, n) V[0][k] = r0[k] / r0_norm;
            nv = 1;
        }
        for (int64_t k = 0; k < (restart + 1) * restart; ++k) h[k] = 0.0;     /* :268 */
        for (int64_t k = 0; k <= restart; ++k) g[k] = 0.0;
        g[0] = r0_norm;                                             /* :270 */
        for (int64_t k = 0; k < restart; ++k) { cs[k] = 0.0; sn[k] = 0.0; }
        int64_t m = 0;
        int happy = 0;
        for (int64_t j = 0; j < restart; ++j) {                     /* :276 */
            iteration += 1;                                         /* :277 */
            if (side == 1) {                                        /* :279-307 */
                kro_spmv(a, V[j], w);
                rc = kro_pc_apply(pc, w, z, n); if (rc) goto out_noupdate;
                mgs2(rs, z, Z, j, h, ld, n);                        /* against Z[0..=j] (:286-298) */
                H(j + 1, j) = kro_norm(rs, z, n);
                if (fabs(H(j + 1, j)) < eps) { happy = 1; break; }  /* :300-303 break BEFORE givens */
                double hj = H(j + 1, j);
                PFOR(k, n) V[nv][k] = z[k] / hj;
                memcpy(Z[nz], V[nv], sizeof(double) * (size_t)n);   /* :305-306 */
                nv++; nz++;
            } else if (side == 2) {                                 /* :308-342 */
                rc = kro_pc_apply(pc, V[j], w, n); if (rc) goto out_noupdate;
                kro_spmv(a, w, z);                                  /* w2 = A w; w2_ortho = clone */
                mgs2(rs, z, V, j, h, ld, n);
                H(j + 1, j) = kro_norm(rs, z, n);
                if (fabs(H(j + 1, j)) < eps) { happy = 1; break; }
                double hj = H(j + 1, j);
                PFOR(k, n) V[nv][k] = z[k] / hj;
                rc = kro_pc_apply(pc, V[nv], Z[nz], n); if (rc) goto out_noupdate;
                nv++; nz++;
            } else if (side == 3) {                                 /* extension: z = M^-1 A v_j, orthogonalised against V */
                kro_spmv(a, V[j], w);
                rc = kro_pc_apply(pc, w, z, n); if (rc) goto out_noupdate;
                mgs2(rs, z, V, j, h, ld, n);
                H(j + 1, j) = kro_norm(rs, z, n);
                if (fabs(H(j + 1, j)) < eps) happy = 1;
                else { double hj = H(j + 1, j); PFOR(k, n) V[nv][k] = z[k] / hj; nv++; }
            } else {                                                /* :343-345 -> arnoldi :65-105 */
                kro_spmv(a, V[j], w);
                mgs2(rs, w, V, j, h, ld, n);
                H(j + 1, j) = kro_norm(rs, w, n);
                if (fabs(H(j + 1, j)) < eps) happy = 1;             /* returns true, NO push, givens still applied */
                else { double hj = H(j + 1, j); PFOR(k, n) V[nv][k] = w[k] / hj; nv++; }
            }
            givens_update(h, ld, g, cs, sn, j, eps);                /* :347 */
            double res_norm = fabs(g[j + 1]);                       /* :348 */
            int stop = conv_check(p->tol, p->max_iters, res_norm, res0_in, iteration, st);   /* :349-350 (res0_in == res0 unless side 3) */
            trace_push(tr, iteration, res_norm);                    /* addition: the reference keeps no GMRES history */
            m = j + 1;                                              /* :351 */
            if ((stop && st->converged) || happy) break;            /* :352-354 */
        }
        back_substitution(h, ld, g, y, m, eps);                     /* :357-360 */
        double** upd = (side == 2) ? Z : V;                         /* :362-386 */
        for (int64_t j = 0; j < m; ++j) {
            const double yj = y[j]; const double* u = upd[j];
            PFOR(k, n) xk[k] = xk[k] + yj * u[k];
        }
        residual(a, b, xk, r0, tmp);                                /* :388-391 */
        beta = kro_norm(rs, r0, n);                                 /* :392 */
        st->final_residual = beta;                                  /* :394 */
        st->converged = beta < p->tol * res0;                       /* :395 */
        if (st->converged || iteration >= p->max_iters) break;      /* :396-398 */
    }
    memcpy(x, xk, sizeof(double) * (size_t)n);                      /* :400 */
out_noupdate:
    for (int64_t k = 0; k <= restart; ++k) { free(V[k]); free(Z[k]); }
    free(V); free(Z); free(xk); free(r0); free(tmp); free(w); free(z);
    free(h); free(g); free(cs); free(sn); free(y);
    return rc;
}
#undef H

/* ------------------------------------------------------------------ CGS (cgs.rs:58-135), as written */
int32_t kro_cgs(const kro_csr_t* a, const kro_pc_t* pc, const double* b, double* x,
                const kro_params_t* p, const kro_reduce_t* rs, kro_stats_t* st, kro_trace_t* tr) {
    (void)pc;                                                               /* :59 */
    const int64_t n = a->nrows;
    double* xk = dalloc(n); double* r = dalloc(n); double* rt = dalloc(n); double* pp = dalloc(n);
    double* q = dzeros(n); double* u = dzeros(n); double* v = dalloc(n); double* upq = dalloc(n); double* w = dalloc(n);
    memcpy(xk, x, sizeof(double) * (size_t)n);                              /* :61 */
    kro_spmv(a, xk, v);                                                     /* :64-69 */
    { PFOR(i, n) r[i] = b[i] - v[i]; }
    memcpy(rt, r, sizeof(double) * (size_t)n);                              /* :70 */
    memcpy(pp, r, sizeof(double) * (size_t)n);                              /* :71 */
    double rho = kro_dot(rs, rt, r, n);                                     /* :74 */
    double rho_old = 0.0;
    const double res0 = kro_norm(rs, r, n);                                 /* :76 */
    st->iterations = 0; st->final_residual = res0; st->converged = 0;       /* :77 */
    for (int64_t i = 1; i <= p->max_iters; ++i) {
        if (fabs(rho) < DBL_EPSILON) break;                                 /* :80-82 */
        if (i == 1) {                                                       /* :83-86 */
            memcpy(u, r, sizeof(double) * (size_t)n);
            memcpy(pp, u, sizeof(double) * (size_t)n);
        } else {                                                            /* :87-99 */
            const double beta = rho / rho_old;
            PFOR(k, n) {
                const double qo = q[k], po = pp[k];
                u[k] = r[k] + beta * qo;
                pp[k] = u[k] + beta * (qo + beta * po);
            }
        }
        kro_spmv(a, pp, v);                                                 /* :101-103 */
        const double alpha = rho / kro_dot(rs, rt, v, n);                   /* :105 */
        { PFOR(k, n) q[k] = u[k] - alpha * v[k]; }                          /* :107-109 */
        { PFOR(k, n) xk[k] += alpha * (u[k] + q[k]); }                      /* :111-113 */
        { PFOR(k, n) upq[k] = u[k] + q[k]; }                                /* :115-118 */
        kro_spmv(a, upq, w);                                                /* :119-120 */
        { PFOR(k, n) r[k] = r[k] - alpha * w[k]; }                          /* :121-123 */
        const double res_norm = kro_norm(rs, r, n);                         /* :124 */
        trace_push(tr, i, res_norm);
        const int stop = conv_check(p->tol, p->max_iters, res_norm, res0, i, st);   /* :126-127 */
        if (stop && st->converged) break;                                   /* :128-131 */
        rho_old = rho;                                                      /* :132-133 */
        rho = kro_dot(rs, rt, r, n);
    }
    memcpy(x, xk, sizeof(double) * (size_t)n);                              /* :129 / :135 */
    free(xk); free(r); free(rt); free(pp); free(q); free(u); free(v); free(upq); free(w);
    return KRO_OK;
}

/* ------------------------------------------------------------------ TFQMR (tfqmr.rs:64-221), as written */
int32_t kro_tfqmr(const kro_csr_t* a, const kro_pc_t* pc, const double* b, double* x,
                  const kro_params_t* p, const kro_reduce_t* rs, kro_stats_t* st, kro_trace_t* tr) {
    (void)pc;
    const int64_t n = a->nrows;
    for (int64_t i = 0; i < n; ++i) x[i] = 0.0;                             /* :72 the initial guess is discarded */
    double* r = dalloc(n); double* rt = dalloc(n);
    memcpy(r, b, sizeof(double) * (size_t)n);                               /* :75 */
    memcpy(rt, r, sizeof(double) * (size_t)n);                              /* :77 */
    double rho = kro_dot(rs, r, rt, n);                                     /* :80 */
    if (rho == 0.0) {                                                       /* :81-83 */
        st->iterations = 0; st->final_residual = kro_norm(rs, r, n); st->converged = 1;
        free(r); free(rt);
        return KRO_OK;
    }
    double* v = dalloc(n); double* w = dalloc(n); double* y = dalloc(n); double* u = dzeros(n); double* d = dzeros(n);
    double* q = dalloc(n); double* t = dalloc(n); double* au = dalloc(n);
    memcpy(w, r, sizeof(double) * (size_t)n); memcpy(y, r, sizeof(double) * (size_t)n);   /* :94-95 */
    double psi_old = 0.0, eta_old = 0.0;
    const double tau = kro_norm(rs, r, n);                                  /* :100 */
    const double res0 = tau;
    st->iterations = 0; st->final_residual = res0; st->converged = 0;       /* :102 */
    int returned = 0;
    if (tau == 0.0) { st->final_residual = 0.0; st->converged = 1; returned = 1; }   /* :103-105 */
    double dpold = tau;                                                     /* :107 */
    for (int64_t k = 1; !returned && k <= p->max_iters; ++k) {
        kro_spmv(a, y, v);                                                  /* :110-112 */
        const double sigma = kro_dot(rs, rt, v, n);                         /* :115 */
        if (sigma == 0.0 || !isfinite(sigma)) {                             /* :116-121 */
            st->final_residual = kro_norm(rs, r, n); st->iterations = k; st->converged = 0; returned = 1; break;
        }
        const double alpha = rho / sigma;                                   /* :122 */
        if (alpha == 0.0 || !isfinite(alpha)) {                             /* :123-128 */
            st->final_residual = kro_norm(rs, r, n); st->iterations = k; st->converged = 0; returned = 1; break;
        }
        { PFOR(i, n) u[i] = r[i] - alpha * v[i]; }                          /* :131-133 */
        { PFOR(i, n) q[i] = u[i] - alpha * v[i]; }                          /* :136-139 */
        { PFOR(i, n) t[i] = u[i] + q[i]; }                                  /* :142-145 */
        kro_spmv(a, t, au);                                                 /* :146-147 */
        { PFOR(i, n) r[i] = r[i] - alpha * au[i]; }                         /* :149-151 */
        const double dp = kro_norm(rs, r, n);                               /* :152 */
        const double tau_m0 = sqrt(dp * dpold);                             /* :153 */
        double tau_local = tau_m0;
        for (int m = 0; m < 2; ++m) {                                       /* :156 */
            const double norm_u_m = (m == 0) ? dp : kro_norm(rs, q, n);     /* :157-161 */
            const double tau_for_m = (m == 0) ? tau_m0 : tau_local;
            const double* u_m = (m == 0) ? u : q;                           /* :162 */
            const double psi = norm_u_m / tau_for_m;                        /* :165 */
            const double c_m = 1.0 / sqrt(1.0 + psi * psi);                 /* :166 */
            const double eta = c_m * c_m * alpha;                           /* :167 */
            const double cf = (alpha == 0.0 || k == 1) ? 0.0 : psi_old * psi_old * eta_old / alpha;   /* :170-174 */
            { PFOR(i, n) d[i] = u_m[i] + cf * d[i]; }                       /* :175-177 */
            { PFOR(i, n) x[i] = x[i] + eta * d[i]; }                        /* :180-182 */
            const double dpest = sqrt((double)(2 * k + m + 2)) * tau_for_m; /* :185 */
            trace_push(tr, k, dpest);
            const int stop = conv_check(p->tol, p->max_iters, dpest, res0, k, st);   /* :186-187 */
            psi_old = psi; eta_old = eta;                                   /* :188-189 */
            tau_local = tau_for_m * psi * c_m;                              /* :190 */
            if (stop) {                                                     /* :191-196 */
                st->final_residual = dpest; st->iterations = k; st->converged = 1; returned = 1; break;
            }
        }
        if (returned) break;
        memcpy(r, u, sizeof(double) * (size_t)n);                           /* :203 */
        const double rho_new = kro_dot(rs, rt, r, n);                       /* :204 */
        const double beta = rho_new / rho;                                  /* :205 */
        rho = rho_new;
        PFOR(i, n) {                                                        /* :208-211 */
            w[i] = u[i] + beta * (q[i] + beta * w[i]);
            y[i] = u[i] + beta * (q[i] + beta * y[i]);
        }
        dpold = dp;                                                         /* :212 */
    }
    if (!returned) { st->final_residual = kro_norm(rs, r, n); st->iterations = p->max_iters; }   /* :215-217 */
    free(r); free(rt); free(v); free(w); free(y); free(u); free(d); free(q); free(t); free(au);
    return KRO_OK;
}

/* ------------------------------------------------------------------ FGMRES (fgmres.rs:114-340), as written */
int32_t kro_fgmres(const kro_csr_t* a, const kro_pc_t* pc, const double* b, double* x,
                   const kro_params_t* p, int32_t orthog, double haptol, int32_t preallocate,
                   const kro_reduce_t* rs, kro_stats_t* st, kro_trace_t* tr) {
    const int64_t n = a->nrows, restart = p->restart, max_iters = p->max_iters;
    const double tol = p->tol;
    const int has_pc = pc && pc->kind != KRO_PC_NONE;
    const int64_t cap = (preallocate ? max_iters : restart) + 1;           /* :144-165 */
    const int64_t ld = preallocate ? max_iters : restart;
    double* r = dalloc(n); double* tmp = dalloc(n); double* w = dalloc(n);
    int32_t rc = KRO_OK;
    memcpy(r, b, sizeof(double) * (size_t)n);                               /* :134-139 */
    kro_spmv(a, x, tmp);
    { PFOR(i, n) r[i] = r[i] - tmp[i]; }
    double beta = kro_norm(rs, r, n);                                       /* :140 */
    if (beta == 0.0) {                                                      /* :141-143 */
        st->iterations = 0; st->final_residual = 0.0; st->converged = 1;
        free(r); free(tmp); free(w);
        return KRO_OK;
    }
    double** V = (double**)calloc((size_t)cap + 1, sizeof(double*)); double** Z = (double**)calloc((size_t)cap + 1, sizeof(double*));
    for (int64_t k = 0; k <= cap; ++k) { V[k] = dzeros(n); Z[k] = dzeros(n); }
    double* h = dzeros((cap + 1) * (ld + 1)); double* cs = dzeros(cap + 1); double* sn = dzeros(cap + 1); double* s = dzeros(cap + 2);
    double* hcol = dzeros(cap + 2); double* y = dzeros(cap + 1);
#define H(i, k) h[(i) * (ld + 1) + (k)]
    s[0] = beta;                                                            /* :166 */
    { PFOR(i, n) V[0][i] = r[i] / beta; }                                   /* :167-169 */
    int64_t total_iters = 0;
    const double res_norm_outer = beta;                                     /* :171 `let res_norm = beta;` (never reassigned) */
    st->iterations = 0; st->final_residual = res_norm_outer; st->converged = 0;
    while (total_iters < max_iters) {                                       /* :175 */
        const int64_t m = preallocate ? (max_iters < restart ? max_iters : restart)
                                      : (restart < max_iters - total_iters ? restart : max_iters - total_iters);   /* :203 */
        int converged = 0;
        int64_t arnoldi_steps = m;
        for (int64_t j = 0; j < m; ++j) {                                   /* :207 */
            memcpy(Z[j], V[j], sizeof(double) * (size_t)n);                 /* :209 */
            if (has_pc) { rc = kro_pc_apply(pc, V[j], Z[j], n); if (rc) goto out; }   /* :210-212 `?` */
            kro_spmv(a, Z[j], w);                                           /* :214-215 */
            for (int64_t i = 0; i <= j; ++i) hcol[i] = kro_dot(rs, w, V[i], n);        /* :220-222 / :231-233 */
            for (int64_t i = 0; i <= j; ++i) { const double hi = hcol[i]; const double* vi = V[i]; PFOR(k, n) w[k] = w[k] - hi * vi[k]; }
            if (orthog == 1)                                                /* :239-247 refinement (h_col is NOT corrected) */
                for (int64_t i = 0; i <= j; ++i) {
                    const double corr = kro_dot(rs, w, V[i], n);
                    if (fabs(corr) > 1e-10) { const double* vi = V[i]; PFOR(k, n) w[k] = w[k] - corr * vi[k]; }
                }
            H(j + 1, j) = kro_norm(rs, w, n);                               /* :250 */
            for (int64_t i = 0; i <= j; ++i) H(i, j) = hcol[i];             /* :251 */
            const double hapbnd = haptol * fabs(s[j]);                      /* :253 */
            const int happy = fabs(H(j + 1, j)) < hapbnd;
            if (!happy) { const double wn = H(j + 1, j); PFOR(k, n) V[j + 1][k] = w[k] / wn; }   /* :255-258 */
            else { PFOR(k, n) V[j + 1][k] = 0.0; }                          /* :259-261 */
            for (int64_t i = 0; i < j; ++i) {                               /* :263-267 */
                const double temp = cs[i] * H(i, j) + sn[i] * H(i + 1, j);
                H(i + 1, j) = -sn[i] * H(i, j) + cs[i] * H(i + 1, j);
                H(i, j) = temp;
            }
            const double h1 = H(j, j), h2 = H(j + 1, j);                    /* :269-278 */
            const double denom = sqrt(h1 * h1 + h2 * h2);
            double c, s_;
            if (denom == 0.0) { c = 1.0; s_ = 0.0; } else { c = h1 / denom; s_ = h2 / denom; }
            cs[j] = c; sn[j] = s_;
            const double temp = c * s[j] + s_ * s[j + 1];                   /* :281-283 */
            s[j + 1] = -s_ * s[j] + c * s[j + 1];
            s[j] = temp;
            H(j, j) = c * H(j, j) + s_ * H(j + 1, j);                       /* :284-285 */
            H(j + 1, j) = 0.0;
            const double res_norm = fabs(s[j + 1]);                         /* :286 */
            total_iters += 1;
            trace_push(tr, total_iters, res_norm);                          /* :289-292 */
            const int stop = conv_check(tol, max_iters, res_norm, s[0], total_iters, st);   /* :293: res0 = the ROTATED s[0] */
            if (stop) {                                                     /* :295-301 */
                st->final_residual = res_norm; st->iterations = total_iters;
                arnoldi_steps = j + 1; converged = 1;
                break;
            }
        }
        const int64_t k = arnoldi_steps;                                    /* :304-314: no zero-pivot guard */
        for (int64_t i = k - 1; i >= 0; --i) {
            double sum = s[i];
            for (int64_t l = i + 1; l < k; ++l) sum = sum - H(i, l) * y[l];
            y[i] = sum / H(i, i);
        }
        for (int64_t i = 0; i < k; ++i) { const double yi = y[i]; const double* zi = Z[i]; PFOR(q, n) x[q] = x[q] + yi * zi[q]; }   /* :348-353 */
        memcpy(r, b, sizeof(double) * (size_t)n);                           /* :317-322 */
        kro_spmv(a, x, tmp);
        { PFOR(i, n) r[i] = r[i] - tmp[i]; }
        const double res_true = kro_norm(rs, r, n);
        if (res_true < tol || converged) {                                  /* :324-329 ABSOLUTE tolerance */
            st->final_residual = res_true; st->iterations = total_iters; st->converged = 1;
            break;
        }
        beta = res_true;                                                    /* :331-337 */
        { PFOR(i, n) V[0][i] = r[i] / beta; }
        for (int64_t q = 0; q <= restart; ++q) s[q] = 0.0;
        s[0] = beta;
    }
    st->final_residual = res_norm_outer;                                    /* :339: the OUTER res_norm == the initial ||r|| */
    st->iterations = total_iters;                                           /* :340 */
#undef H
out:
    for (int64_t k = 0; k <= cap; ++k) { free(V[k]); free(Z[k]); }
    free(V); free(Z); free(h); free(cs); free(sn); free(s); free(hcol); free(y); free(r); free(tmp); free(w);
    return rc;
}

/* ------------------------------------------------------------------ BiCGStab (bicgstab.rs:69-293) */
static int32_t bicgstab_impl(const kro_csr_t* a, const kro_pc_t* pc, const double* b, double* x,
                             const kro_params_t* p, const kro_reduce_t* rs, kro_stats_t* st, kro_trace_t* tr,
                             int use_pc) {
    const int64_t n = a->nrows;
    const double EPS = DBL_EPSILON;                                 /* T::epsilon() */
    double* xk = dalloc(n); memcpy(xk, x, sizeof(double) * (size_t)n);
    double* r = dalloc(n); double* rhat = dalloc(n); double* v = dzeros(n); double* pp = dalloc(n);
    double* s = dalloc(n); double* t = dalloc(n); double* tmp = dalloc(n);
    double* ph = use_pc ? dalloc(n) : NULL; double* sh = use_pc ? dalloc(n) : NULL;
    int32_t rc = KRO_OK;
    residual(a, b, xk, r, tmp);                                     /* :75-77 */
    memcpy(rhat, r, sizeof(double) * (size_t)n);                    /* :78 */
    double rho_prev = 1.0, alpha = 1.0, omega_prev = 1.0;           /* :79-81 */
    memcpy(pp, r, sizeof(double) * (size_t)n);                      /* :83 */
    double res0 = kro_norm(rs, r, n);                               /* :85-96 */
    st->iterations = 0; st->final_residual = res0; st->converged = 0;
    trace_push(tr, 0, res0);                                        /* addition: no history in the reference */
    if (res0 <= p->tol) { st->converged = 1; goto out; }            /* :98-102 ABSOLUTE tolerance */
    for (int64_t i = 1; i <= p->max_iters; ++i) {                   /* :103 */
        double rho = kro_dot(rs, rhat, r, n);                       /* :105-116 */
        if (fabs(rho) < EPS) break;                                 /* :117-119 */
        double beta = (i == 1) ? 0.0 : (rho / rho_prev) * (alpha / omega_prev);    /* :120-124 */
        PFOR(j, n) pp[j] = r[j] + beta * (pp[j] - omega_prev * v[j]);              /* :134 / :140 */
        if (use_pc) { rc = kro_pc_apply(pc, pp, ph, n); if (rc) goto out_noupdate; kro_spmv(a, ph, v); }
        else kro_spmv(a, pp, v);                                    /* :144-146 */
        double alpha_den = kro_dot(rs, rhat, v, n);                 /* :149-160 */
        if (fabs(alpha_den) < EPS) break;                           /* :161-163 */
        alpha = rho / alpha_den;                                    /* :164 */
        PFOR(j, n) s[j] = r[j] - alpha * v[j];                      /* :166-175 */
        double s_norm = kro_norm(rs, s, n);                         /* :177-188 */
        if (s_norm <= p->tol) {                                     /* :189-206 */
            const double* dir = use_pc ? ph : pp;
            PFOR(j, n) xk[j] = xk[j] + alpha * dir[j];
            st->iterations = i; st->final_residual = s_norm; st->converged = 1;
            trace_push(tr, i, s_norm);
            goto out;
        }
        if (use_pc) { rc = kro_pc_apply(pc, s, sh, n); if (rc) goto out_noupdate; kro_spmv(a, sh, t); }
        else kro_spmv(a, s, t);                                     /* :208-209 */
        double omega_num = kro_dot(rs, t, s, n);                    /* :211-222 */
        double omega_den = kro_dot(rs, t, t, n);                    /* :223-234 */
        if (fabs(omega_den) < EPS) break;                           /* :235-237 */
        double omega = omega_num / omega_den;                       /* :238 */
        { const double* d1 = use_pc ? ph : pp; const double* d2 = use_pc ? sh : s;
          PFOR(j, n) xk[j] = xk[j] + alpha * d1[j] + omega * d2[j]; }              /* :246 / :252 */
        PFOR(j, n) r[j] = s[j] - omega * t[j];                      /* :256-266 */
        double r_norm = kro_norm(rs, r, n);                         /* :268-279 */
        st->iterations = i; st->final_residual = r_norm; st->converged = (r_norm <= p->tol);   /* :280 */
        trace_push(tr, i, r_norm);
        if (r_norm <= p->tol) goto out;                             /* :281-284 */
        if (fabs(omega) < EPS) break;                               /* :285-287 */
        rho_prev = rho; omega_prev = omega;                         /* :288-289 */
    }
out:
    memcpy(x, xk, sizeof(double) * (size_t)n);                      /* :291 */
out_noupdate:
    free(xk); free(r); free(rhat); free(v); free(pp); free(s); free(t); free(tmp); free(ph); free(sh);
    return rc;
}

int32_t kro_bicgstab(const kro_csr_t* a, const kro_pc_t* pc, const double* b, double* x,
                     const kro_params_t* p, const kro_reduce_t* rs, kro_stats_t* st, kro_trace_t* tr) {
    (void)pc;                                                       /* bicgstab.rs:70 `let _ = pc;` */
    return bicgstab_impl(a, NULL, b, x, p, rs, st, tr, 0);
}

/* EXTENSION: right-preconditioned BiCGStab (p^ = M^-1 p, v = A p^; s^ = M^-1 s, t = A s^; x += alpha p^ + omega s^),
 * everything else exactly as bicgstab.rs.  With pc == None it degenerates to kro_bicgstab. */
int32_t kro_bicgstab_rpc(const kro_csr_t* a, const kro_pc_t* pc, const double* b, double* x,
                     const kro_params_t* p, const kro_reduce_t* rs, kro_stats_t* st, kro_trace_t* tr) {
    int use_pc = pc && pc->kind != KRO_PC_NONE;
    return bicgstab_impl(a, pc, b, x, p, rs, st, tr, use_pc);
}
