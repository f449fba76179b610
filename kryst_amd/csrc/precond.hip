// Pointwise preconditioner applies for gfx950: Jacobi (src/preconditioner/jacobi.rs), the Chebyshev filter
// (src/preconditioner/chebyshev.rs:83-140) and the ILU(0)-family triangular solves (ilu.rs / ilup.rs; in ilu.hip).
#include "pc.h"
#include "ew.h"
#include <cfloat>
#include <algorithm>
#include <cmath>

namespace kr {

// ---------------------------------------------------------------- Jacobi
// jacobi.rs:53-73 builds diag[i] = (A e_i)[i] with n matvecs; the row sum that lands in diag[i] is
// 0 + a_ii*1 (+ exact zeros) = a_ii, so reading the stored diagonal gives the same bits.
__global__ void jacobi_setup_kernel(const int32_t* row_ptr, const int32_t* col, const double* val, int32_t nrows,
                                    double* inv_diag) {
    const int32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nrows) return;
    double d = 0.0;
    for (int32_t k = row_ptr[i]; k < row_ptr[i + 1]; ++k)
        if (col[k] == i) d = 0.0 + val[k] * 1.0;
    inv_diag[i] = (d != 0.0) ? 1.0 / d : 0.0;                       // jacobi.rs:69-71
}

struct JacobiOp {                    // y[i] = inv_diag[i] * x[i]   (jacobi.rs:84-92)
    static constexpr int NQ = 0; static constexpr const char* TAG = "Jacobi";
    const double* inv; const double* x; double* y;
    __device__ __forceinline__ void pair(int64_t i, bool, bool, double (&)[1]) const {
        const d2 a = ld2(inv, i), b = ld2(x, i);
        st2(y, i, a.a * b.a, a.b * b.b);
    }
};

// ---------------------------------------------------------------- Chebyshev filter
struct Cheb1Op {                     // v1[i] = (v1[i] - c*v0[i]) / d            (chebyshev.rs:105-107)
    static constexpr int NQ = 0; static constexpr const char* TAG = "Cheb1";
    double c, d; const double* v0; double* v1;
    __device__ __forceinline__ void pair(int64_t i, bool, bool, double (&)[1]) const {
        const d2 a = ld2(v0, i), b = ld2(v1, i);
        st2(v1, i, (b.a - c * a.a) / d, (b.b - c * a.b) / d);
    }
};
struct Cheb2Op {                     // v2[i] = (2*(v2[i] - c*v1[i]) / d) - v0[i] (chebyshev.rs:121)
    static constexpr int NQ = 0; static constexpr const char* TAG = "Cheb2";
    double c, d; const double* v0; const double* v1; double* v2;
    __device__ __forceinline__ void pair(int64_t i, bool, bool, double (&)[1]) const {
        const d2 a = ld2(v0, i), b = ld2(v1, i), w = ld2(v2, i);
        st2(v2, i, (2.0 * (w.a - c * b.a) / d) - a.a, (2.0 * (w.b - c * b.b) / d) - a.b);
    }
};
struct Cheb2ScaleOp {                // the last recurrence step and z[i] = tau * v2[i] (chebyshev.rs:121 then :130-138) in one pass
    static constexpr int NQ = 0; static constexpr const char* TAG = "Cheb2Scale";
    double c, d, tau; const double* v0; const double* v1; const double* v2; double* z;
    __device__ __forceinline__ void pair(int64_t i, bool, bool, double (&)[1]) const {
        const d2 a = ld2(v0, i), b = ld2(v1, i), w = ld2(v2, i);
        const double n0 = (2.0 * (w.a - c * b.a) / d) - a.a, n1 = (2.0 * (w.b - c * b.b) / d) - a.b;   // v2 as the reference stores it
        st2(z, i, tau * n0, tau * n1);
    }
};
struct ScaleOp {                     // z[i] = tau * v[i]                         (chebyshev.rs:130-138)
    static constexpr int NQ = 0; static constexpr const char* TAG = "Scale";
    double tau; const double* v; double* z;
    __device__ __forceinline__ void pair(int64_t i, bool, bool, double (&)[1]) const {
        const d2 a = ld2(v, i);
        st2(z, i, tau * a.a, tau * a.b);
    }
};

static double chebyshev_t_host(int64_t m, double x) {       // chebyshev.rs:143-159
    if (m == 0) return 1.0;
    if (m == 1) return x;
    double t0 = 1.0, t1 = x, t2;
    for (int64_t k = 2; k <= m; ++k) { t2 = 2.0 * x * t1 - t0; t0 = t1; t1 = t2; }
    return t1;
}

static inline int64_t padded(int64_t n) { return (n + KR_TILE - 1) / KR_TILE * KR_TILE; }

int32_t chebyshev_dev(kryst_csr_t a, const double* r, double* z, double alpha, double beta, int64_t m,
                      double* v0, double* v1, double* v2, const int* done) {
    kryst_ctx_t ctx = a->ctx;
    const int64_t n = a->nrows;
    const size_t bytes = sizeof(double) * (size_t)padded(n);
    if (std::fabs(beta - alpha) < DBL_EPSILON) {                                           // :88-92
        KR_HIP(hipMemcpyAsync(z, r, bytes, hipMemcpyDeviceToDevice, ctx->s_main));
        return KRYST_OK;
    }
    const double c = (beta + alpha) / 2.0;
    const double d = (beta - alpha) / 2.0;
    const double tau = 1.0 / chebyshev_t_host(m, (0.0 - c) / d);                           // :102
    // v0 = r is never written by the recurrence, so r itself plays v0 (no copy); the three work vectors rotate behind it
    const double* p0 = r;                                                                  // :103 v0 = r.clone()
    double* bufs[3] = {v0, v1, v2};
    double* p1 = bufs[1];
    KR_TRY(launch_spmv(a, p0, p1, 0, nullptr, done));                                      // :104
    KR_TRY(launch_ew(ctx, Cheb1Op{c, d, p0, p1}, n, done));
    if (m == 0) { KR_HIP(hipMemcpyAsync(z, r, bytes, hipMemcpyDeviceToDevice, ctx->s_main)); return KRYST_OK; }
    if (m == 1) { KR_HIP(hipMemcpyAsync(z, p1, bytes, hipMemcpyDeviceToDevice, ctx->s_main)); return KRYST_OK; }   // unscaled
    int next = 2;                                                                          // index of the free buffer
    for (int64_t k = 2; k <= m; ++k) {
        double* p2 = bufs[next];
        KR_TRY(launch_spmv(a, p1, p2, 0, nullptr, done));
        if (k == m) return launch_ew(ctx, Cheb2ScaleOp{c, d, tau, p0, p1, p2, z}, n, done);   // :121 + :130-138
        KR_TRY(launch_ew(ctx, Cheb2Op{c, d, p0, p1, p2}, n, done));
        // swap(v0,v1); swap(v1,v2): the old v0 buffer becomes the free one (r is never recycled)
        const double* old0 = p0;
        p0 = p1; p1 = p2;
        next = (old0 == r) ? 0 : (int)(std::find(bufs, bufs + 3, old0) - bufs);
    }
    return KRYST_OK;
}

int32_t ilu_apply_dev(kryst_pc_t pc, const double* r, double* z, const int* done);   // ilu.hip
void    ilu_free(kryst_pc_t pc);
int32_t ilu_health(kryst_pc_t pc);
bool    ilu_fell_back(kryst_pc_t pc);
bool    ilu_is_wavefront(kryst_pc_t pc);
int32_t pc_health(kryst_pc_t pc) { return (pc && pc->kind == KR_PC_ILU && pc->d_work) ? ilu_health(pc) : KRYST_OK; }
bool pc_fell_back(kryst_pc_t pc) { return pc && pc->kind == KR_PC_ILU && pc->d_work && ilu_fell_back(pc); }

static int32_t pc_apply_kind(kryst_pc_t pc, const double* r, double* z, const int* done) {
    kryst_ctx_t ctx = pc->ctx;
    switch (pc->kind) {
        case KR_PC_IDENTITY:
            if (r != z) KR_HIP(hipMemcpyAsync(z, r, sizeof(double) * (size_t)padded(pc->n), hipMemcpyDeviceToDevice, ctx->s_main));
            return KRYST_OK;
        case KR_PC_JACOBI: return launch_ew(ctx, JacobiOp{pc->d_inv_diag, r, z}, pc->n, done);
        case KR_PC_ILU: return ilu_apply_dev(pc, r, z, done);
        case KR_PC_CHEB_STUB:
            set_error("Chebyshev preconditioner requires matrix argument; use apply_chebyshev free function.");   // chebyshev.rs:69
            return KRYST_SOLVE_ERROR;
        case KR_PC_CHEB:
            return chebyshev_dev(pc->a, r, z, pc->cheb_alpha, pc->cheb_beta, pc->cheb_degree, pc->d_v0, pc->d_v1, pc->d_v2, done);
        case KR_PC_SPAI:      // ApproxInv::apply (approxinv.rs:268-298): z_i = sum_j M_ij r_j, ascending j from 0 -- an SpMV with M
            return launch_spmv(pc->a, r, z, 0, nullptr, done);
        default: set_error("unknown preconditioner kind %d", pc->kind); return KRYST_UNSUPPORTED;
    }
}
int32_t pc_apply_dev(kryst_pc_t pc, const double* r, double* z, const int* done) {
    const int32_t rc = pc_apply_kind(pc, r, z, done);
    phase_mark(pc->ctx, KR_PH_PC);
    return rc;
}

}  // namespace kr

using namespace kr;

// all device work of a context is ordered on ctx->s_main (a non-blocking stream: the null stream does NOT order with it)
static int32_t alloc_vec(kryst_ctx_t ctx, double** p, int64_t n) {
    const size_t bytes = sizeof(double) * (size_t)(padded(n) + KR_TILE);
    KR_HIP(hipMalloc(p, bytes));
    KR_HIP(hipMemsetAsync(*p, 0, bytes, ctx->s_main));
    return KRYST_OK;
}

extern "C" {

int32_t kryst_pc_identity(kryst_ctx_t ctx, kryst_pc_t* out) {
    KR_ARG(ctx && out, "pc_identity");
    kryst_pc_t pc = new kryst_pc_s();
    pc->ctx = ctx; pc->kind = KR_PC_IDENTITY; pc->n = -1;
    *out = pc;
    return KRYST_OK;
}

int32_t kryst_pc_jacobi(kryst_csr_t a, kryst_pc_t* out) {
    KR_ARG(a && out, "pc_jacobi");
    KR_HIP(hipSetDevice(a->ctx->device));
    kryst_pc_t pc = new kryst_pc_s();
    pc->ctx = a->ctx; pc->kind = KR_PC_JACOBI; pc->a = a; pc->n = a->nrows;
    int32_t rc = alloc_vec(a->ctx, &pc->d_inv_diag, pc->n);
    if (rc != KRYST_OK) { delete pc; return rc; }
    if (pc->n > 0) {
        hipLaunchKernelGGL(jacobi_setup_kernel, dim3((unsigned)((pc->n + 255) / 256)), dim3(256), 0, a->ctx->s_main,
                           a->d_row_ptr, a->d_col, a->d_val, (int32_t)pc->n, pc->d_inv_diag);
        KR_HIP(hipGetLastError());
    }
    *out = pc;
    return KRYST_OK;
}

int32_t kryst_pc_chebyshev_stub(kryst_ctx_t ctx, int32_t degree, kryst_pc_t* out) {
    KR_ARG(ctx && out, "pc_chebyshev_stub");
    kryst_pc_t pc = new kryst_pc_s();
    pc->ctx = ctx; pc->kind = KR_PC_CHEB_STUB; pc->cheb_degree = degree; pc->n = -1;
    *out = pc;
    return KRYST_OK;
}

int32_t kryst_pc_chebyshev(kryst_csr_t a, double alpha, double beta, int32_t degree, kryst_pc_t* out) {
    KR_ARG(a && out && degree >= 0, "pc_chebyshev");
    KR_ARG(a->nrows == a->xlen, "pc_chebyshev: square operator required");
    KR_HIP(hipSetDevice(a->ctx->device));
    kryst_pc_t pc = new kryst_pc_s();
    pc->ctx = a->ctx; pc->kind = KR_PC_CHEB; pc->a = a; pc->n = a->nrows;
    pc->cheb_alpha = alpha; pc->cheb_beta = beta; pc->cheb_degree = degree;
    int32_t rc = alloc_vec(a->ctx, &pc->d_v0, pc->n);
    if (rc == KRYST_OK) rc = alloc_vec(a->ctx, &pc->d_v1, pc->n);
    if (rc == KRYST_OK) rc = alloc_vec(a->ctx, &pc->d_v2, pc->n);
    if (rc != KRYST_OK) { kryst_pc_destroy(pc); return rc; }
    *out = pc;
    return KRYST_OK;
}

int32_t kryst_pc_apply(kryst_pc_t pc, kryst_vec_t r, kryst_vec_t z) {
    KR_ARG(pc && r && z, "pc_apply");
    KR_ARG(r->ctx == pc->ctx && z->ctx == pc->ctx, "pc_apply: context mismatch");
    KR_ARG(r->n == z->n, "pc_apply: length mismatch");
    KR_ARG(pc->n < 0 || pc->n == r->n, "pc_apply: vector length != operator size");
    KR_HIP(hipSetDevice(pc->ctx->device));
    if (pc->n < 0) {                                     // identity / stub carry no size
        kryst_pc_s tmp = *pc; tmp.n = r->n;
        return pc_apply_dev(&tmp, r->d, z->d, nullptr);
    }
    KR_TRY(pc_apply_dev(pc, r->d, z->d, nullptr));
    if (pc->kind == KR_PC_ILU && pc->d_work && ilu_is_wavefront(pc)) {
        // the wavefront solve relies on in-order workgroup dispatch (tri_wave.h): wait, and if it gave up repeat the apply
        // with the plane kernels (same bits)
        KR_HIP(hipStreamSynchronize(pc->ctx->s_main));
        if (pc_health(pc) != KRYST_OK) {
            (void)pc_fell_back(pc);
            KR_TRY(pc_apply_dev(pc, r->d, z->d, nullptr));
        }
    }
    return KRYST_OK;
}

// measurement hook (bench.py): `reps` back-to-back applies between two HIP events on the compute stream, one warm-up apply
// before them (it also captures the ILU apply's graph); the wavefront solve's health is checked once at the end
int32_t kryst_bench_pc_apply(kryst_pc_t pc, kryst_vec_t r, kryst_vec_t z, int32_t reps, double* avg_ms) {
    KR_ARG(pc && r && z && avg_ms && reps >= 1, "bench_pc_apply");
    KR_ARG(r->ctx == pc->ctx && z->ctx == pc->ctx && r->n == z->n && pc->n == r->n, "bench_pc_apply: size or context mismatch");
    kryst_ctx_t ctx = pc->ctx;
    KR_HIP(hipSetDevice(ctx->device));
    KR_TRY(pc_apply_dev(pc, r->d, z->d, nullptr));
    KR_HIP(hipStreamSynchronize(ctx->s_main));
    if (pc_health(pc) != KRYST_OK) { (void)pc_fell_back(pc); }          // measure what the preconditioner now runs
    KR_HIP(hipEventRecord(ctx->tm0, ctx->s_main));
    for (int k = 0; k < reps; ++k) KR_TRY(pc_apply_dev(pc, r->d, z->d, nullptr));
    KR_HIP(hipEventRecord(ctx->tm1, ctx->s_main));
    KR_HIP(hipEventSynchronize(ctx->tm1));
    float ms = 0.f;
    KR_HIP(hipEventElapsedTime(&ms, ctx->tm0, ctx->tm1));
    *avg_ms = (double)ms / reps;
    KR_HIP(hipStreamSynchronize(ctx->s_main));
    return pc_health(pc);
}

int32_t kryst_pc_approx_inverse(kryst_csr_t m, kryst_pc_t* out) {
    KR_ARG(m && out, "pc_approx_inverse");
    KR_ARG(m->nrows == m->xlen, "pc_approx_inverse: the inverse rows must form a square operator");
    kryst_pc_t pc = new kryst_pc_s();
    pc->ctx = m->ctx; pc->kind = KR_PC_SPAI; pc->a = m; pc->n = m->nrows;
    *out = pc;
    return KRYST_OK;
}

int32_t kryst_pc_destroy(kryst_pc_t pc) {
    if (!pc) return KRYST_OK;
    (void)hipSetDevice(pc->ctx->device);
    (void)hipStreamSynchronize(pc->ctx->s_main);
    (void)hipFree(pc->d_inv_diag); (void)hipFree(pc->d_v0); (void)hipFree(pc->d_v1); (void)hipFree(pc->d_v2);
    ilu_free(pc);
    delete pc;
    return KRYST_OK;
}

int32_t kryst_apply_chebyshev(kryst_csr_t a, kryst_vec_t r, kryst_vec_t z, double alpha, double beta, int64_t m) {
    KR_ARG(a && r && z && m >= 0, "apply_chebyshev");
    KR_ARG(r->n == a->nrows && z->n == a->nrows && a->nrows == a->xlen, "apply_chebyshev: size mismatch");
    KR_HIP(hipSetDevice(a->ctx->device));
    double *v0 = nullptr, *v1 = nullptr, *v2 = nullptr;
    int32_t rc = alloc_vec(a->ctx, &v0, a->nrows);
    if (rc == KRYST_OK) rc = alloc_vec(a->ctx, &v1, a->nrows);
    if (rc == KRYST_OK) rc = alloc_vec(a->ctx, &v2, a->nrows);
    if (rc == KRYST_OK) rc = chebyshev_dev(a, r->d, z->d, alpha, beta, m, v0, v1, v2, nullptr);
    (void)hipStreamSynchronize(a->ctx->s_main);
    (void)hipFree(v0); (void)hipFree(v1); (void)hipFree(v2);
    return rc;
}

}  // extern "C"
