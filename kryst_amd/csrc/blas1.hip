// Device vectors and BLAS-1: InnerProduct for () (src/core/wrappers.rs:90-127) and the pointwise loops the
// reference solvers run on Vec<f64>.  All kernels are HBM-bound streams (16 B per lane per array).
#include "ew.h"
#include "dist.h"

namespace kr {

// ---------------------------------------------------------------- ops
struct DotOp {                       // wrappers.rs:90-108: sum of x[i]*y[i]
    static constexpr int NQ = 1; static constexpr const char* TAG = "Dot"; static constexpr int BPC = 4;
    const double* x; const double* y;
    __device__ __forceinline__ void pair(int64_t i, bool in0, bool in1, double (&acc)[1]) const {
        const d2 a = ld2(x, i), b = ld2(y, i);
        if (in0) acc[0] = acc[0] + a.a * b.a;
        if (in1) acc[0] = acc[0] + a.b * b.b;
    }
};
struct AxpyOp {                      // y[i] = y[i] + alpha*x[i]   (cg.rs:207-209)
    static constexpr int NQ = 0; static constexpr const char* TAG = "Axpy";
    Coef alpha; const double* x; double* y;
    __device__ __forceinline__ void pair(int64_t i, bool, bool, double (&)[1]) const {
        const double al = alpha.get();
        const d2 a = ld2(x, i), b = ld2(y, i);
        st2(y, i, b.a + al * a.a, b.b + al * a.b);
    }
};
struct AypxOp {                      // y[i] = x[i] + beta*y[i]    (cg.rs:274-276)
    static constexpr int NQ = 0; static constexpr const char* TAG = "Aypx";
    Coef beta; const double* x; double* y;
    __device__ __forceinline__ void pair(int64_t i, bool, bool, double (&)[1]) const {
        const double be = beta.get();
        const d2 a = ld2(x, i), b = ld2(y, i);
        st2(y, i, a.a + be * b.a, a.b + be * b.b);
    }
};
struct SubOp {                       // out[i] = a[i] - b[i]       (cg.rs:123 `bi - ax`)
    static constexpr int NQ = 0; static constexpr const char* TAG = "Sub";
    const double* a; const double* b; double* out;
    __device__ __forceinline__ void pair(int64_t i, bool, bool, double (&)[1]) const {
        const d2 u = ld2(a, i), v = ld2(b, i);
        st2(out, i, u.a - v.a, u.b - v.b);
    }
};
struct FillOp {
    static constexpr int NQ = 0; static constexpr const char* TAG = "Fill";
    double v; double* out;
    __device__ __forceinline__ void pair(int64_t i, bool in0, bool in1, double (&)[1]) const {
        st2(out, i, in0 ? v : 0.0, in1 ? v : 0.0);       // padding stays zero
    }
};
struct SplitmixOp {                  // SURVEY 8d synthetic data
    static constexpr int NQ = 0; static constexpr const char* TAG = "Splitmix";
    uint64_t seed; int64_t goff; double* out;
    __device__ __forceinline__ static double gen(uint64_t seed, uint64_t idx) {
        uint64_t z = seed + (idx + 1) * 0x9E3779B97F4A7C15ull;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        z = z ^ (z >> 31);
        return (double)(z >> 11) * (1.0 / 9007199254740992.0);
    }
    __device__ __forceinline__ void pair(int64_t i, bool in0, bool in1, double (&)[1]) const {
        st2(out, i, in0 ? gen(seed, (uint64_t)(goff + i)) : 0.0, in1 ? gen(seed, (uint64_t)(goff + i + 1)) : 0.0);
    }
};

// ---------------------------------------------------------------- final fold
template <int NQ>
__global__ __launch_bounds__(KR_F) void final_fold_kernel(const double* partials, int64_t stride, int64_t ntiles,
                                                          double* chunks, int64_t cstride, unsigned int* ticket, unsigned int* err, double* out) {
    __shared__ double lds[NQ * (KR_F / 64)];
    double v[NQ];
    const int f = fold2<NQ>(partials, stride, ntiles, chunks, cstride, ticket, err, v, lds);
    if (!f) return;
    if (threadIdx.x == 0) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) out[q] = f == 2 ? __longlong_as_double(0x7FF8000000000000ll) : v[q];      // (gave up: *err is raised, the host asks fold_gave_up)
    }
}

bool fold_gave_up(kryst_ctx_t ctx) {
    unsigned int w[2] = {0, 0};          // [0]: a fold's polling hand-off, [1]: a halo pull (spmv.hip: halo_pull_kernel) ran out of patience
    if (hipMemcpyAsync(w, fold_err(ctx), sizeof w, hipMemcpyDeviceToHost, ctx->s_main) != hipSuccess || hipStreamSynchronize(ctx->s_main) != hipSuccess) { (void)hipGetLastError(); return false; }
    if (!w[0] && !w[1]) return false;
    (void)hipMemsetAsync(fold_err(ctx), 0, sizeof w, ctx->s_main);
    (void)hipStreamSynchronize(ctx->s_main);
    if (w[1]) {
        set_error("halo exchange by peer stores: a neighbour's epoch stamp never arrived (poll budget exhausted); the halo was poisoned with NaNs -- "
                  "switch the operator back with kryst_csr_halo_mode(a, 0)");
        return true;
    }
    ctx->fold_poll_off = true;
    set_error("an inner product's two-level fold gave up waiting for a workgroup of its own launch (GPU shared, time-sliced or serialised by a "
              "profiler for seconds); the result was discarded and this context now uses the ticket hand-off");
    return true;
}

// A one-rank communicator normally skips RCCL; KRYST_FORCE_COMM=1 keeps the collective path (used by the
// single-GPU rehearsal of the multi-rank code in tests/test_gpu_y_dist_single.py).
bool use_collectives(kryst_ctx_t ctx) {
    if (ctx->nranks > 1) return true;
    static int force = -1;
    if (force < 0) { const char* e = getenv("KRYST_FORCE_COMM"); force = (e && atoi(e) != 0) ? 1 : 0; }
    return force == 1 && ctx->comm != nullptr;
}

__global__ void chunk_arm_kernel(double* chunks, int64_t count) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) chunks[i] = __longlong_as_double((long long)KR_FOLD_UNSET);
}

int32_t ensure_partials(kryst_ctx_t ctx, int64_t ntiles) {
    if (ntiles <= ctx->partials_cap) return KRYST_OK;
    if (ctx->d_partials) { KR_HIP(hipStreamSynchronize(ctx->s_main)); KR_HIP(hipFree(ctx->d_partials)); ctx->d_partials = nullptr; }
    int64_t cap = ntiles + 64;
    KR_HIP(hipMalloc(&ctx->d_partials, sizeof(double) * (size_t)cap * KR_MAXQ));
    ctx->partials_cap = cap;
    if (ctx->d_chunks) { KR_HIP(hipFree(ctx->d_chunks)); ctx->d_chunks = nullptr; }
    ctx->chunks_cap = nchunks_of(cap) + 8;
    KR_HIP(hipMalloc(&ctx->d_chunks, sizeof(double) * (size_t)ctx->chunks_cap * KR_MAXQ));
    // every cell "unset" (fold2): a NaN payload, written on the compute stream in front of the first fold
    hipLaunchKernelGGL(chunk_arm_kernel, dim3((unsigned)((ctx->chunks_cap * KR_MAXQ + 255) / 256)), dim3(256), 0, ctx->s_main, ctx->d_chunks, ctx->chunks_cap * KR_MAXQ);
    KR_HIP(hipGetLastError());
    return KRYST_OK;
}

int32_t launch_dot_partials(kryst_ctx_t ctx, const double* x, const double* y, int64_t n, int slot) {
    (void)slot;
    return launch_ew(ctx, DotOp{x, y}, n);
}

int32_t launch_final_fold(kryst_ctx_t ctx, int nq, int64_t ntiles, double* d_out) {
    KR_TRY(ensure_partials(ctx, ntiles > 0 ? ntiles : 1));
    switch (nq) {
        case 1: hipLaunchKernelGGL(final_fold_kernel<1>, dim3((unsigned)nchunks_of(ntiles)), dim3(KR_F), 0, ctx->s_main, ctx->d_partials, ctx->partials_cap, ntiles, ctx->d_chunks, ctx->chunks_cap, fold_ticket(ctx), fold_err(ctx), d_out); break;
        case 2: hipLaunchKernelGGL(final_fold_kernel<2>, dim3((unsigned)nchunks_of(ntiles)), dim3(KR_F), 0, ctx->s_main, ctx->d_partials, ctx->partials_cap, ntiles, ctx->d_chunks, ctx->chunks_cap, fold_ticket(ctx), fold_err(ctx), d_out); break;
        case 3: hipLaunchKernelGGL(final_fold_kernel<3>, dim3((unsigned)nchunks_of(ntiles)), dim3(KR_F), 0, ctx->s_main, ctx->d_partials, ctx->partials_cap, ntiles, ctx->d_chunks, ctx->chunks_cap, fold_ticket(ctx), fold_err(ctx), d_out); break;
        case 4: hipLaunchKernelGGL(final_fold_kernel<4>, dim3((unsigned)nchunks_of(ntiles)), dim3(KR_F), 0, ctx->s_main, ctx->d_partials, ctx->partials_cap, ntiles, ctx->d_chunks, ctx->chunks_cap, fold_ticket(ctx), fold_err(ctx), d_out); break;
        case 8: hipLaunchKernelGGL(final_fold_kernel<8>, dim3((unsigned)nchunks_of(ntiles)), dim3(KR_F), 0, ctx->s_main, ctx->d_partials, ctx->partials_cap, ntiles, ctx->d_chunks, ctx->chunks_cap, fold_ticket(ctx), fold_err(ctx), d_out); break;
        default: set_error("final fold: nq=%d", nq); return KRYST_ERR_ARG;
    }
    KR_HIP(hipGetLastError());
    return KRYST_OK;
}

int32_t vec_check2(kryst_vec_t a, kryst_vec_t b) {
    KR_ARG(a && b, "null vector");
    KR_ARG(a->ctx == b->ctx, "vectors belong to different contexts");
    KR_ARG(a->n == b->n, "vector length mismatch");
    return KRYST_OK;
}

}  // namespace kr

using namespace kr;

static inline int64_t padded(int64_t n) { return (n + KR_TILE - 1) / KR_TILE * KR_TILE; }

extern "C" {

int32_t kryst_vec_create(kryst_ctx_t ctx, int64_t n, kryst_vec_t* out) {
    KR_ARG(ctx && out && n >= 0, "vec_create");
    KR_HIP(hipSetDevice(ctx->device));
    kryst_vec_t v = new kryst_vec_s();
    v->ctx = ctx; v->n = n;
    const size_t bytes = sizeof(double) * (size_t)(padded(n) + KR_TILE);
    if (hipMalloc(&v->d, bytes) != hipSuccess) { delete v; set_error("hipMalloc(%zu) failed", bytes); return KRYST_ERR_HIP; }
    KR_HIP(hipMemsetAsync(v->d, 0, bytes, ctx->s_main));
    *out = v;
    return KRYST_OK;
}

int32_t kryst_vec_destroy(kryst_vec_t v) {
    if (!v) return KRYST_OK;
    (void)hipSetDevice(v->ctx->device);
    (void)hipStreamSynchronize(v->ctx->s_main);
    (void)hipFree(v->d);
    delete v;
    return KRYST_OK;
}

int32_t kryst_vec_len(kryst_vec_t v, int64_t* n) { KR_ARG(v && n, "vec_len"); *n = v->n; return KRYST_OK; }

int32_t kryst_vec_upload(kryst_vec_t v, const double* host, int64_t n) {
    KR_ARG(v && host, "vec_upload");
    KR_ARG(n == v->n, "vec_upload: length mismatch");
    KR_HIP(hipSetDevice(v->ctx->device));
    KR_HIP(hipMemcpyAsync(v->d, host, sizeof(double) * (size_t)n, hipMemcpyHostToDevice, v->ctx->s_main));
    KR_HIP(hipStreamSynchronize(v->ctx->s_main));
    return KRYST_OK;
}

int32_t kryst_vec_download(kryst_vec_t v, double* host, int64_t n) {
    KR_ARG(v && host, "vec_download");
    KR_ARG(n == v->n, "vec_download: length mismatch");
    KR_HIP(hipSetDevice(v->ctx->device));
    KR_HIP(hipMemcpyAsync(host, v->d, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost, v->ctx->s_main));
    KR_HIP(hipStreamSynchronize(v->ctx->s_main));
    return KRYST_OK;
}

int32_t kryst_vec_fill(kryst_vec_t v, double value) {
    KR_ARG(v, "vec_fill");
    KR_HIP(hipSetDevice(v->ctx->device));
    return launch_ew(v->ctx, FillOp{value, v->d}, v->n);
}

int32_t kryst_vec_copy(kryst_vec_t dst, kryst_vec_t src) {
    KR_TRY(vec_check2(dst, src));
    KR_HIP(hipSetDevice(dst->ctx->device));
    KR_HIP(hipMemcpyAsync(dst->d, src->d, sizeof(double) * (size_t)padded(src->n), hipMemcpyDeviceToDevice, dst->ctx->s_main));
    return KRYST_OK;
}

int32_t kryst_vec_fill_splitmix(kryst_vec_t v, uint64_t seed, int64_t global_offset) {
    KR_ARG(v, "vec_fill_splitmix");
    KR_HIP(hipSetDevice(v->ctx->device));
    return launch_ew(v->ctx, SplitmixOp{seed, global_offset, v->d}, v->n);
}

int32_t kryst_dot(kryst_vec_t x, kryst_vec_t y, double* out) {
    KR_TRY(vec_check2(x, y));
    KR_ARG(out, "dot: out");
    kryst_ctx_t ctx = x->ctx;
    KR_HIP(hipSetDevice(ctx->device));
    KR_TRY(launch_ew(ctx, DotOp{x->d, y->d}, x->n));
    // result slot of its own (d_scal[1024]): d_scal[0..512) is the scalar state of an open solve / stepping session, and the
    // partials / chunk / ticket scratch is consumed in stream order, so a dot between two session steps disturbs nothing
    double* slot = ctx->d_scal + 1024;
    if (!use_collectives(ctx)) {
        KR_TRY(launch_final_fold(ctx, 1, ntiles_of(x->n), slot));
        KR_HIP(hipMemcpyAsync(ctx->h_pinned, slot, sizeof(double), hipMemcpyDeviceToHost, ctx->s_main));
        KR_HIP(hipStreamSynchronize(ctx->s_main));
        *out = ctx->h_pinned[0];
        if (*out != *out && fold_gave_up(ctx)) return KRYST_ERR_HIP;
        return KRYST_OK;
    }
    // several ranks: local fold -> all-gather of one double per rank -> the result goes to the host anyway, so the rank-ordered
    // fold (total = r0; total = total + r_p: the same bits on every rank) runs there: two launches and one collective per dot,
    // like the solvers' reduce_then (whose rank-ordered fold shares a launch with the scalar step that consumes it)
    double* gathered = ctx->d_scal + 1032;                               // nranks doubles (<= 2048 ranks)
    KR_ARG(ctx->nranks <= 2048, "dot: too many ranks for the scalar scratch");
    KR_TRY(launch_final_fold(ctx, 1, ntiles_of(x->n), slot));
    KR_TRY(comm_all_gather(ctx, slot, gathered, 1));
    KR_HIP(hipMemcpyAsync(ctx->h_pinned, gathered, sizeof(double) * (size_t)ctx->nranks, hipMemcpyDeviceToHost, ctx->s_main));
    KR_HIP(hipStreamSynchronize(ctx->s_main));
    double total = ctx->h_pinned[0];
    for (int p = 1; p < ctx->nranks; ++p) total = total + ctx->h_pinned[p];
    *out = total;
    if (total != total && fold_gave_up(ctx)) return KRYST_ERR_HIP;
    return KRYST_OK;
}

int32_t kryst_norm(kryst_vec_t x, double* out) {
    double d = 0.0;
    KR_TRY(kryst_dot(x, x, &d));
    *out = __builtin_sqrt(d);      // wrappers.rs:126 `.sqrt()` (IEEE correctly rounded on the host)
    return KRYST_OK;
}

int32_t kryst_axpy(double alpha, kryst_vec_t x, kryst_vec_t y) {
    KR_TRY(vec_check2(x, y));
    KR_HIP(hipSetDevice(x->ctx->device));
    return launch_ew(x->ctx, AxpyOp{coef_val(alpha), x->d, y->d}, x->n);
}

int32_t kryst_aypx(double beta, kryst_vec_t x, kryst_vec_t y) {
    KR_TRY(vec_check2(x, y));
    KR_HIP(hipSetDevice(x->ctx->device));
    return launch_ew(x->ctx, AypxOp{coef_val(beta), x->d, y->d}, x->n);
}

int32_t kryst_sub(kryst_vec_t a, kryst_vec_t b, kryst_vec_t out) {
    KR_TRY(vec_check2(a, b));
    KR_TRY(vec_check2(a, out));
    KR_HIP(hipSetDevice(a->ctx->device));
    return launch_ew(a->ctx, SubOp{a->d, b->d, out->d}, a->n);
}

}  // extern "C"
