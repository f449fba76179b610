// Measurement-only entry point: times the library's three stream shapes (Gram-Schmidt link, batched dots, CG
// update) on vectors carved from ONE pool at a chosen byte stride, so that the effect of the vectors' relative
// placement in HBM can be measured in isolation (tools/stride_bench.py).
#include "solver_common.h"

namespace kr {
struct BenchLinkOp {                 // z = z - h b0 ; partial (z, b1)      -- 3 reads, 1 write
    static constexpr int NQ = 1;
    double h; const double* b0; const double* b1; double* z;
    __device__ __forceinline__ void pair(int64_t i, bool in0, bool in1, double (&acc)[1]) const {
        const d2 zz = ld2(z, i), b = ld2(b0, i), nx = ld2(b1, i);
        const double z0 = zz.a - h * b.a, z1 = zz.b - h * b.b;
        st2(z, i, z0, z1);
        if (in0) acc[0] = acc[0] + z0 * nx.a;
        if (in1) acc[0] = acc[0] + z1 * nx.b;
    }
};
struct BenchDot8Op {                 // 8 dots against one w               -- 9 reads
    static constexpr int NQ = 8; static constexpr int BPC = 4;
    const double* w; const double* v[8];
    __device__ __forceinline__ void pair(int64_t i, bool in0, bool in1, double (&acc)[8]) const {
        const d2 ww = ld2(w, i);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const d2 vv = ld2(v[k], i);
            if (in0) acc[k] = acc[k] + ww.a * vv.a;
            if (in1) acc[k] = acc[k] + ww.b * vv.b;
        }
    }
};
struct BenchCgOp {                   // x += a p ; r -= a q ; partial (r,r) -- 4 reads, 2 writes
    static constexpr int NQ = 1;
    double al; const double* p; const double* ap; double* x; double* r;
    __device__ __forceinline__ void pair(int64_t i, bool in0, bool in1, double (&acc)[1]) const {
        const d2 pp = ld2(p, i), aa = ld2(ap, i), xx = ld2(x, i), rr = ld2(r, i);
        const double r0 = rr.a - al * aa.a, r1 = rr.b - al * aa.b;
        st2(x, i, xx.a + al * pp.a, xx.b + al * pp.b); st2(r, i, r0, r1);
        if (in0) acc[0] = acc[0] + r0 * r0;
        if (in1) acc[0] = acc[0] + r1 * r1;
    }
};
typedef double v2dd __attribute__((ext_vector_type(2)));
template <bool NTL, bool NTS>
struct BenchCgNtOp {                 // BenchCgOp with nontemporal loads and/or stores
    static constexpr int NQ = 1;
    double al; const double* p; const double* ap; double* x; double* r;
    __device__ __forceinline__ v2dd L(const double* q, int64_t i) const {
        if constexpr (NTL) return __builtin_nontemporal_load(reinterpret_cast<const v2dd*>(q + i));
        else return *reinterpret_cast<const v2dd*>(q + i);
    }
    __device__ __forceinline__ void S(double* q, int64_t i, double a, double b) const {
        v2dd v; v.x = a; v.y = b;
        if constexpr (NTS) __builtin_nontemporal_store(v, reinterpret_cast<v2dd*>(q + i));
        else *reinterpret_cast<v2dd*>(q + i) = v;
    }
    __device__ __forceinline__ void pair(int64_t i, bool in0, bool in1, double (&acc)[1]) const {
        const v2dd pp = L(p, i), aa = L(ap, i), xx = L(x, i), rr = L(r, i);
        const double r0 = rr.x - al * aa.x, r1 = rr.y - al * aa.y;
        S(x, i, xx.x + al * pp.x, xx.y + al * pp.y); S(r, i, r0, r1);
        if (in0) acc[0] = acc[0] + r0 * r0;
        if (in1) acc[0] = acc[0] + r1 * r1;
    }
};
struct BenchAypxOp {                 // p = r + beta p                      -- 2 reads, 1 write (CG's direction update)
    static constexpr int NQ = 0;
    double be; const double* x; double* y;
    __device__ __forceinline__ void pair(int64_t i, bool, bool, double (&)[1]) const {
        const d2 a = ld2(x, i), b = ld2(y, i);
        st2(y, i, a.a + be * b.a, a.b + be * b.b);
    }
};
struct BenchCgResOp {                // r -= a q ; partial (r,r)            -- 2 reads, 1 write (CG's residual pass, x update deferred)
    static constexpr int NQ = 1; static constexpr int BPC = 3;
    double al; const double* ap; double* r;
    __device__ __forceinline__ void pair(int64_t i, bool in0, bool in1, double (&acc)[1]) const {
        const d2 aa = ld2(ap, i), rr = ld2(r, i);
        const double r0 = rr.a - al * aa.a, r1 = rr.b - al * aa.b;
        st2(r, i, r0, r1);
        if (in0) acc[0] = acc[0] + r0 * r0;
        if (in1) acc[0] = acc[0] + r1 * r1;
    }
};
struct BenchCgDirOp {                // x += a p ; p = r + b p             -- 3 reads, 2 writes (CG's direction pass with the deferred x update)
    static constexpr int NQ = 0; static constexpr int BPC = 3;       // as CgDirectionOp (3 per CU is what pays INSIDE the iteration; alone, 2 is 10 % faster)
    double al, be; const double* r; double* p; double* x;
    __device__ __forceinline__ void pair(int64_t i, bool, bool, double (&)[1]) const {
        const d2 pp = ld2(p, i), xx = ld2(x, i), rr = ld2(r, i);
        st2(x, i, xx.a + al * pp.a, xx.b + al * pp.b);
        st2(p, i, rr.a + be * pp.a, rr.b + be * pp.b);
    }
};
}  // namespace kr

using namespace kr;

extern "C" int32_t kryst_bench_streams(kryst_ctx_t ctx, int64_t n, int64_t stride_bytes, int32_t kind, int32_t reps, double* avg_ms) {
    KR_ARG(ctx && avg_ms && n > 0 && reps >= 1 && kind >= 0 && kind <= 8, "bench_streams");
    const int nvec = kind == 0 ? 3 : kind == 1 ? 9 : (kind == 6 || kind == 7) ? 2 : kind == 8 ? 3 : 4;
    const size_t vbytes = padded_bytes(n) + sizeof(double) * KR_TILE;
    KR_ARG(stride_bytes % 16 == 0 && (size_t)stride_bytes >= vbytes, "bench_streams: stride must be a multiple of 16 and hold a padded vector");
    KR_HIP(hipSetDevice(ctx->device));
    char* pool = nullptr;
    KR_HIP(hipMalloc(&pool, (size_t)stride_bytes * nvec));
    KR_HIP(hipMemsetAsync(pool, 0, (size_t)stride_bytes * nvec, ctx->s_main));
    double* v[9];
    for (int k = 0; k < nvec; ++k) v[k] = reinterpret_cast<double*>(pool + (size_t)k * stride_bytes);
    int32_t rc = KRYST_OK;
    auto once = [&]() -> int32_t {
        if (kind == 0) return launch_ew(ctx, BenchLinkOp{0.5, v[1], v[2], v[0]}, n);
        if (kind == 1) { BenchDot8Op op; op.w = v[0]; for (int k = 0; k < 8; ++k) op.v[k] = v[k + 1]; return launch_ew(ctx, op, n); }
        if (kind == 6) return launch_ew(ctx, BenchAypxOp{0.5, v[0], v[1]}, n);
        if (kind == 7) return launch_ew(ctx, BenchCgResOp{0.5, v[0], v[1]}, n);
        if (kind == 8) return launch_ew(ctx, BenchCgDirOp{1e-3, 0.5, v[0], v[1], v[2]}, n);
        if (kind == 3) return launch_ew(ctx, BenchCgNtOp<true, true>{0.5, v[0], v[1], v[2], v[3]}, n);
        if (kind == 4) return launch_ew(ctx, BenchCgNtOp<false, true>{0.5, v[0], v[1], v[2], v[3]}, n);
        if (kind == 5) return launch_ew(ctx, BenchCgNtOp<true, false>{0.5, v[0], v[1], v[2], v[3]}, n);
        return launch_ew(ctx, BenchCgOp{0.5, v[0], v[1], v[2], v[3]}, n);
    };
    rc = once();
    if (rc == KRYST_OK) {
        (void)hipEventRecord(ctx->tm0, ctx->s_main);
        for (int r = 0; r < reps && rc == KRYST_OK; ++r) rc = once();
        (void)hipEventRecord(ctx->tm1, ctx->s_main);
        (void)hipEventSynchronize(ctx->tm1);
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, ctx->tm0, ctx->tm1);
        *avg_ms = (double)ms / reps;
    }
    (void)hipStreamSynchronize(ctx->s_main);
    (void)hipFree(pool);
    return rc;
}

// ---- the plain-CSR SpMV's traffic WITHOUT its arithmetic (bench.py: `stream_skeleton`): per 512-row tile a workgroup reads the tile's row
// pointers (8 bytes per lane; the stream bounds come out of them, as in spmv_wave_kernel), streams the tile's values (16 bytes per lane and
// request) and column indices (8 bytes per lane and request), four requests of each in flight per wave, reads x once (16 bytes per lane) and
// writes y once (16 bytes per lane, nontemporal) -- SURVEY 8(d)'s bytes, 12 nnz + 4 (n + 1) + 16 n, on the operator's OWN arrays (same
// allocation, same placement), one tile per workgroup in index order.  No gathers, no products, no row sums, no fold: what a kernel with this
// traffic mix reaches on this HBM, measured in the same process as the real kernel.  y receives garbage.
namespace kr {
typedef unsigned int sk_u4 __attribute__((ext_vector_type(4)));
typedef unsigned int sk_u2 __attribute__((ext_vector_type(2)));
template <bool NT>
__global__ __launch_bounds__(KR_T) void csr_skeleton_kernel(const int32_t* __restrict__ rp, const int32_t* __restrict__ col, const double* __restrict__ val,
                                                            const double* __restrict__ x, double* __restrict__ y, int64_t n, int64_t ntiles) {
    const int t = threadIdx.x, l = t & 63, w = t >> 6;
    const int64_t q = blockIdx.x;
    if (q >= ntiles) return;
    const int64_t r0 = q * KR_TILE, r1 = r0 + KR_TILE < n ? r0 + KR_TILE : n;
    const int64_t ri = r0 + 2 * t < n ? r0 + 2 * t : ((n - 1) & ~(int64_t)1);
    const sk_u2 mine = *reinterpret_cast<const sk_u2*>(rp + ri);              // the lane's two row pointers (row_ptr holds n + 1 + 8 entries)
    unsigned acc = mine.x ^ mine.y;
    const int64_t k0 = (int64_t)rp[r0] & ~(int64_t)1, k1 = rp[r1];            // (uniform: scalar loads) the tile's entries, from an even index
    const int64_t wq = (((k1 - k0 + 3) / 4) + 1) & ~(int64_t)1;              // a contiguous quarter per wave, an even number of entries
    const int64_t e0 = k0 + w * wq, e1 = e0 + wq < k1 ? e0 + wq : k1;
    for (int64_t off = e0; off < e1; off += 4 * 128) {
        sk_u4 v[4]; sk_u2 c[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t k = off + u * 128 + 2 * l;                          // (val / col carry 8 entries of padding: k + 1 <= nnz + 7)
            v[u] = sk_u4{0u, 0u, 0u, 0u}; c[u] = sk_u2{0u, 0u};
            if (k < e1) {
                const sk_u4* pv = reinterpret_cast<const sk_u4*>(val + k);
                const sk_u2* pc = reinterpret_cast<const sk_u2*>(col + k);
                v[u] = NT ? __builtin_nontemporal_load(pv) : *pv;
                c[u] = NT ? __builtin_nontemporal_load(pc) : *pc;
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) acc ^= v[u].x ^ v[u].y ^ v[u].z ^ v[u].w ^ c[u].x ^ c[u].y;
    }
    const sk_u4 xv = *reinterpret_cast<const sk_u4*>(x + r0 + 2 * t);         // vectors are padded to whole tiles
    acc ^= xv.x ^ xv.w;
    sk_u4 yv; yv.x = acc; yv.y = (unsigned)t; yv.z = 0u; yv.w = 1u;
    __builtin_nontemporal_store(yv, reinterpret_cast<sk_u4*>(y + r0 + 2 * t));
}
}  // namespace kr

// ---- EXPERIMENT (round 5): the same bytes as ONE stream.  The plain kernel's skeleton reads two matrix streams that advance at different rates
// (8 B and 4 B per entry) plus row pointers, x and y; a read-only stream of the same size runs 10-20 % faster on this HBM.  Here the entries are
// addressed in CHUNKS of 256: chunk c = 256 column indices (1 KiB) followed by 256 values (2 KiB), 3 KiB apart -- one sequential stream of
// 12 bytes per entry.  Traffic only (the buffer's contents are not looked at); `packed` must hold 3072 * ceil((nnz + 8) / 256) bytes.
namespace kr {
template <bool NT>
__global__ __launch_bounds__(KR_T) void csr_skeleton_packed_kernel(const int32_t* __restrict__ rp, const char* __restrict__ packed,
                                                                   const double* __restrict__ x, double* __restrict__ y, int64_t n, int64_t ntiles) {
    const int t = threadIdx.x, l = t & 63, w = t >> 6;
    const int64_t q = blockIdx.x;
    if (q >= ntiles) return;
    const int64_t r0 = q * KR_TILE, r1 = r0 + KR_TILE < n ? r0 + KR_TILE : n;
    const int64_t ri = r0 + 2 * t < n ? r0 + 2 * t : ((n - 1) & ~(int64_t)1);
    const sk_u2 mine = *reinterpret_cast<const sk_u2*>(rp + ri);
    unsigned acc = mine.x ^ mine.y;
    const int64_t k0 = (int64_t)rp[r0] & ~(int64_t)1, k1 = rp[r1];
    const int64_t wq = (((k1 - k0 + 3) / 4) + 1) & ~(int64_t)1;
    const int64_t e0 = k0 + w * wq, e1 = e0 + wq < k1 ? e0 + wq : k1;
    for (int64_t off = e0; off < e1; off += 4 * 128) {
        sk_u4 v[4]; sk_u2 c[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t k = off + u * 128 + 2 * l;
            v[u] = sk_u4{0u, 0u, 0u, 0u}; c[u] = sk_u2{0u, 0u};
            if (k < e1) {
                const char* chunk = packed + (k >> 8) * 3072;
                const sk_u2* pc = reinterpret_cast<const sk_u2*>(chunk + (k & 255) * 4);
                const sk_u4* pv = reinterpret_cast<const sk_u4*>(chunk + 1024 + (k & 255) * 8);
                v[u] = NT ? __builtin_nontemporal_load(pv) : *pv;
                c[u] = NT ? __builtin_nontemporal_load(pc) : *pc;
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) acc ^= v[u].x ^ v[u].y ^ v[u].z ^ v[u].w ^ c[u].x ^ c[u].y;
    }
    const sk_u4 xv = *reinterpret_cast<const sk_u4*>(x + r0 + 2 * t);
    acc ^= xv.x ^ xv.w;
    sk_u4 yv; yv.x = acc; yv.y = (unsigned)t; yv.z = 0u; yv.w = 1u;
    __builtin_nontemporal_store(yv, reinterpret_cast<sk_u4*>(y + r0 + 2 * t));
}
}  // namespace kr
extern "C" int32_t kryst_debug_csr_skeleton_packed(kryst_csr_t a, kryst_vec_t x, kryst_vec_t y, int32_t reps, double* avg_ms) {
    KR_ARG(a && x && y && avg_ms && reps >= 1 && !a->dist && a->nrows == a->xlen && a->d_row_ptr, "debug_csr_skeleton_packed");
    kryst_ctx_t ctx = a->ctx;
    KR_HIP(hipSetDevice(ctx->device));
    const int64_t nt = ntiles_of(a->nrows);
    char* packed = nullptr;
    const size_t bytes = (size_t)3072 * (size_t)((a->nnz + 8 + 255) / 256 + 1);
    KR_HIP(hipMalloc(&packed, bytes));
    KR_HIP(hipMemsetAsync(packed, 0, bytes, ctx->s_main));
    const bool nontemporal = a->nrows * 8 > (256ll << 20);
    auto once = [&] {
        if (nontemporal) hipLaunchKernelGGL((csr_skeleton_packed_kernel<true>), dim3((unsigned)nt), dim3(KR_T), 0, ctx->s_main, a->d_row_ptr, packed, x->d, y->d, a->nrows, nt);
        else hipLaunchKernelGGL((csr_skeleton_packed_kernel<false>), dim3((unsigned)nt), dim3(KR_T), 0, ctx->s_main, a->d_row_ptr, packed, x->d, y->d, a->nrows, nt);
    };
    once();
    (void)hipEventRecord(ctx->tm0, ctx->s_main);
    for (int r = 0; r < reps; ++r) once();
    (void)hipEventRecord(ctx->tm1, ctx->s_main);
    (void)hipEventSynchronize(ctx->tm1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, ctx->tm0, ctx->tm1);
    *avg_ms = (double)ms / reps;
    (void)hipStreamSynchronize(ctx->s_main);
    (void)hipFree(packed);
    KR_HIP(hipGetLastError());
    return KRYST_OK;
}

// ---- where the CSR arrays live (KRYST_CSR_PLACEMENT_TRIES, csr.h: csr_place).  Round 4 measured that the SAME plain-CSR stream mix runs at 0.70 to
// 0.76 of peak depending on where the driver put the operator's three arrays (profiles/r04/plain_csr_placement_with_skeleton_512.txt: six
// instances in one process, the skeleton itself 2.28 .. 2.50 ms at 512^3) -- a property of the allocation, not of the kernel.  With K > 1 the
// creation allocates up to K - 1 further homes for (row_ptr, col, val), copies the arrays over, times the traffic skeleton on each home and
// keeps the fastest; the others go back to the driver.  Both homes are alive while they are compared (the allocator would hand the old one
// straight back otherwise), so the try is skipped when the device has no room for a second copy.
namespace kr {
static int32_t skeleton_ms(kryst_ctx_t ctx, const int32_t* rp, const int32_t* col, const double* val, const double* x, double* y, int64_t n, bool nt, int reps, double* out) {
    const int64_t ntl = ntiles_of(n);
    auto once = [&] {
        if (nt) hipLaunchKernelGGL((csr_skeleton_kernel<true>), dim3((unsigned)ntl), dim3(KR_T), 0, ctx->s_main, rp, col, val, x, y, n, ntl);
        else hipLaunchKernelGGL((csr_skeleton_kernel<false>), dim3((unsigned)ntl), dim3(KR_T), 0, ctx->s_main, rp, col, val, x, y, n, ntl);
    };
    once();
    KR_HIP(hipGetLastError());
    double best = 1e300;
    for (int round = 0; round < 3; ++round) {                                   // the fastest of three short batches: one disturbed batch must not decide
        (void)hipEventRecord(ctx->tm0, ctx->s_main);
        for (int r = 0; r < reps; ++r) once();
        (void)hipEventRecord(ctx->tm1, ctx->s_main);
        (void)hipEventSynchronize(ctx->tm1);
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, ctx->tm0, ctx->tm1);
        best = std::min(best, (double)ms / reps);
    }
    KR_HIP(hipGetLastError());
    *out = best;
    return KRYST_OK;
}

int32_t csr_place(kryst_csr_t a) {
    kryst_ctx_t ctx = a->ctx;
    a->placement_tries = 1; a->placement_chosen = 0;
    const int64_t n = a->nrows;
    const size_t b_rp = sizeof(int32_t) * (size_t)(n + 1 + 8), b_col = sizeof(int32_t) * (size_t)(a->nnz + 8), b_val = sizeof(double) * (size_t)(a->nnz + 8);
    // default: three homes for operators whose CSR arrays exceed 4 GB (where the spread was measured), one otherwise
    const int K = std::min(8, std::max(1, env_int("KRYST_CSR_PLACEMENT_TRIES", (b_col + b_val) > ((size_t)4 << 30) ? 3 : 1)));
    if (K <= 1 || a->dist || n != a->xlen || n < KR_TILE || !a->d_row_ptr || !a->d_col || !a->d_val) return KRYST_OK;
    const size_t vb = sizeof(double) * (size_t)((n + KR_TILE - 1) / KR_TILE * KR_TILE + KR_TILE);
    double *x = nullptr, *y = nullptr;
    if (hipMalloc(&x, vb) != hipSuccess || hipMalloc(&y, vb) != hipSuccess) { (void)hipGetLastError(); (void)hipFree(x); (void)hipFree(y); return KRYST_OK; }
    (void)hipMemsetAsync(x, 0, vb, ctx->s_main);
    const bool nt = n * 8 > (256ll << 20);
    const int reps = 4;
    int32_t rc = skeleton_ms(ctx, a->d_row_ptr, a->d_col, a->d_val, x, y, n, nt, reps, &a->placement_ms[0]);
    double best = a->placement_ms[0];
    for (int k = 1; k < K && rc == KRYST_OK; ++k) {
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || free_b < (b_rp + b_col + b_val) + ((size_t)8 << 30)) { (void)hipGetLastError(); break; }
        int32_t* rp2 = nullptr; int32_t* col2 = nullptr; double* val2 = nullptr;
        if (hipMalloc(&val2, b_val) != hipSuccess || hipMalloc(&col2, b_col) != hipSuccess || hipMalloc(&rp2, b_rp) != hipSuccess) {
            (void)hipGetLastError(); (void)hipFree(val2); (void)hipFree(col2); (void)hipFree(rp2); break;
        }
        if (hipMemcpyAsync(rp2, a->d_row_ptr, b_rp, hipMemcpyDeviceToDevice, ctx->s_main) != hipSuccess ||
            hipMemcpyAsync(col2, a->d_col, b_col, hipMemcpyDeviceToDevice, ctx->s_main) != hipSuccess ||
            hipMemcpyAsync(val2, a->d_val, b_val, hipMemcpyDeviceToDevice, ctx->s_main) != hipSuccess) { (void)hipGetLastError(); (void)hipStreamSynchronize(ctx->s_main); (void)hipFree(val2); (void)hipFree(col2); (void)hipFree(rp2); rc = KRYST_ERR_HIP; break; }
        double ms = 0.0;
        rc = skeleton_ms(ctx, rp2, col2, val2, x, y, n, nt, reps, &ms);
        a->placement_ms[k] = ms; a->placement_tries = k + 1;
        (void)hipStreamSynchronize(ctx->s_main);
        if (rc == KRYST_OK && ms < best) {                                       // the new home is faster: the old one goes back to the driver
            best = ms; a->placement_chosen = k;
            (void)hipFree(a->d_row_ptr); (void)hipFree(a->d_col); (void)hipFree(a->d_val);
            a->d_row_ptr = rp2; a->d_col = col2; a->d_val = val2;
        } else {
            (void)hipFree(rp2); (void)hipFree(col2); (void)hipFree(val2);
        }
    }
    (void)hipStreamSynchronize(ctx->s_main);
    (void)hipFree(x); (void)hipFree(y);
    return rc;
}
}  // namespace kr

// tries made, the home kept and the traffic skeleton's milliseconds on each home tried (at most 8), see csr_place above
extern "C" int32_t kryst_csr_placement_info(kryst_csr_t a, int32_t* tries, int32_t* chosen, double* skeleton_ms8) {
    KR_ARG(a, "csr_placement_info");
    if (tries) *tries = a->placement_tries;
    if (chosen) *chosen = a->placement_chosen;
    if (skeleton_ms8) for (int k = 0; k < 8; ++k) skeleton_ms8[k] = k < a->placement_tries ? a->placement_ms[k] : 0.0;
    return KRYST_OK;
}

namespace kr {
__global__ __launch_bounds__(256) void poison_lds_kernel(int words) {
    extern __shared__ __attribute__((aligned(16))) unsigned long long poison_smem[];
    for (int i = threadIdx.x; i < words; i += blockDim.x) poison_smem[i] = 0x7ff8badc0ffee000ull + (unsigned)i;
    __syncthreads();
    if (poison_smem[(threadIdx.x * 31) % words] == 1) __builtin_trap();       // (keeps the stores)
}
}  // namespace kr

extern "C" int32_t kryst_bench_poison_lds(kryst_ctx_t ctx) {
    KR_ARG(ctx, "bench_poison_lds");
    KR_HIP(hipSetDevice(ctx->device));
    const int bytes = 152 * 1024;                                               // (160 KiB per CU: one such workgroup per CU at a time)
    KR_HIP(hipFuncSetAttribute((const void*)kr::poison_lds_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    for (int pass = 0; pass < 2; ++pass) hipLaunchKernelGGL(kr::poison_lds_kernel, dim3(2048), dim3(256), bytes, ctx->s_main, bytes / 8);
    KR_HIP(hipGetLastError());
    KR_HIP(hipStreamSynchronize(ctx->s_main));
    return KRYST_OK;
}

extern "C" int32_t kryst_bench_csr_skeleton(kryst_csr_t a, kryst_vec_t x, kryst_vec_t y, int32_t reps, double* avg_ms) {
    KR_ARG(a && x && y && avg_ms && reps >= 1, "bench_csr_skeleton");
    KR_ARG(x->n == a->xlen && y->n == a->nrows && a->nrows == a->xlen && !a->dist, "bench_csr_skeleton: a square single-rank operator and vectors of its size");
    KR_ARG(a->nrows >= 1 && a->d_row_ptr && a->d_col && a->d_val, "bench_csr_skeleton: the operator keeps no CSR arrays");
    kryst_ctx_t ctx = a->ctx;
    KR_HIP(hipSetDevice(ctx->device));
    const int64_t nt = ntiles_of(a->nrows);
    const bool nontemporal = a->nrows * 8 > (256ll << 20);                       // as spmv_wave_kernel: matrix streams bypass the caches beyond the Infinity Cache
    auto once = [&] {
        if (nontemporal) hipLaunchKernelGGL((csr_skeleton_kernel<true>), dim3((unsigned)nt), dim3(KR_T), 0, ctx->s_main, a->d_row_ptr, a->d_col, a->d_val, x->d, y->d, a->nrows, nt);
        else hipLaunchKernelGGL((csr_skeleton_kernel<false>), dim3((unsigned)nt), dim3(KR_T), 0, ctx->s_main, a->d_row_ptr, a->d_col, a->d_val, x->d, y->d, a->nrows, nt);
    };
    once();
    KR_HIP(hipGetLastError());
    (void)hipEventRecord(ctx->tm0, ctx->s_main);
    for (int r = 0; r < reps; ++r) once();
    (void)hipEventRecord(ctx->tm1, ctx->s_main);
    (void)hipEventSynchronize(ctx->tm1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, ctx->tm0, ctx->tm1);
    *avg_ms = (double)ms / reps;
    KR_HIP(hipGetLastError());
    KR_HIP(hipStreamSynchronize(ctx->s_main));
    return KRYST_OK;
}

// ---- EXPERIMENT (round 5): the plain-CSR traffic as a PERSISTENT, SOFTWARE-PIPELINED stream.  The BLAS-1 kernels reach 0.77 of peak with a capped
// grid whose workgroups stride over the vector (a narrow moving window keeps DRAM pages open; ew.h), the one-tile-per-workgroup skeleton 0.70-0.72,
// and a persistent SpMV grid WITHOUT prefetch 0.51-0.57 (each tile is a chain of three dependent round trips).  Here ONE workgroup per CU strides
// over the tiles and keeps the matrix streams of the next TWO tiles in flight by LDS-DMA (global_load_lds: no registers, 12 requests of 1 KiB /
// 256 B per wave and tile into a ring of three tile buffers), row pointers three tiles ahead (scalar loads), x one tile ahead; every wait is a
// counted s_waitcnt.  Traffic only -- the buffers are xor-ed into y.  Sized for tiles of at most 4 x 896 entries (7 per row).
namespace kr {
constexpr int SKP_VAL = 7, SKP_C4 = 3, SKP_C1 = 2;                          // requests per wave and tile: 7 x 1 KiB of values, 3 x 1 KiB + 2 x 256 B of columns
constexpr int SKP_WAVE_BYTES = SKP_VAL * 1024 + SKP_C4 * 1024 + SKP_C1 * 256;   // 10 752
constexpr int SKP_REQ = SKP_VAL + SKP_C4 + SKP_C1;                           // 12
template <int AUX>
__global__ __launch_bounds__(KR_T) void csr_skeleton_persist_kernel(const int32_t* __restrict__ rp, const int32_t* __restrict__ col, const double* __restrict__ val,
                                                                    const double* __restrict__ x, double* __restrict__ y, int64_t n, int64_t ntiles, int64_t nnz) {
    extern __shared__ __attribute__((aligned(16))) unsigned char skp_lds[];
    const int t = threadIdx.x, l = t & 63;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);
    const int64_t G = gridDim.x;
    const int64_t trips = (ntiles + G - 1) / G;
    const int64_t kmax = (nnz + 6) & ~(int64_t)3;                              // last entry a 16-byte request may start at (the arrays carry 8 entries of padding)
    auto tile_of = [&](int64_t j) -> int64_t { const int64_t q = blockIdx.x + j * G; return q < ntiles ? q : ntiles - 1; };
    // the row pointers of a tile are LOADED in one trip (two scalar loads, issued together) and USED in the next: no trip waits for them
    auto rows_of = [&](int64_t q, int32_t& ka, int32_t& kb) {
        const int64_t r0 = q * KR_TILE, r1 = r0 + KR_TILE < n ? r0 + KR_TILE : n;
        ka = rp[r0]; kb = rp[r1];
    };
    auto quarter = [&](int32_t ka, int32_t kb) -> int64_t {                    // first entry of this wave's quarter of the tile
        const int64_t k0 = (int64_t)ka & ~(int64_t)3, k1 = kb;
        const int64_t wq = (((k1 - k0 + 3) / 4) + 3) & ~(int64_t)3;
        return k0 + w * wq;
    };
    auto request = [&](int64_t e0, int slot) {                                 // the 12 requests of one tile into ring slot `slot`
        unsigned char* base = skp_lds + ((size_t)slot * 4 + w) * SKP_WAVE_BYTES;
#pragma unroll
        for (int u = 0; u < SKP_VAL; ++u) {
            int64_t k = e0 + u * 128 + 2 * l; k = k < kmax ? k : kmax;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(val + k), (__attribute__((address_space(3))) void*)(base + u * 1024), 16, 0, AUX & 3);
        }
#pragma unroll
        for (int u = 0; u < SKP_C4; ++u) {
            int64_t k = e0 + u * 256 + 4 * l; k = k < kmax ? k : kmax;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(col + k), (__attribute__((address_space(3))) void*)(base + SKP_VAL * 1024 + u * 1024), 16, 0, AUX & 3);
        }
#pragma unroll
        for (int u = 0; u < SKP_C1; ++u) {
            int64_t k = e0 + SKP_C4 * 256 + u * 64 + l; k = k < kmax ? k : kmax;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(col + k), (__attribute__((address_space(3))) void*)(base + (SKP_VAL + SKP_C4) * 1024 + u * 256), 4, 0, AUX & 3);
        }
    };
    // row pointers by VECTOR loads issued by hand (every lane the same address; a scalar load's wait is the compiler's, and it puts it right behind the
    // load): those of tile j + 4 are requested in trip j, and trip j + 2 -- two counted waits later -- turns them into the tile's requests
    auto rows_req = [&](int64_t q, int32_t& va, int32_t& vb) {
        const int64_t r0 = q * KR_TILE, r1 = r0 + KR_TILE < n ? r0 + KR_TILE : n;
        asm volatile("global_load_dword %0, %1, off" : "=v"(va) : "v"(rp + r0) : "memory");
        asm volatile("global_load_dword %0, %1, off" : "=v"(vb) : "v"(rp + r1) : "memory");
    };
    int32_t ka, kb;
    rows_of(tile_of(0), ka, kb); request(quarter(ka, kb), 0);
    rows_of(tile_of(1), ka, kb); request(quarter(ka, kb), 1);
    rows_of(tile_of(2), ka, kb);                                              // (tile 2: used by the first trip)
    int32_t a3, b3, a4, b4;
    rows_req(tile_of(3), a3, b3);
    sk_u4 xv, xn; sk_u2 pv, pn;                                               // x and the lane's own two row pointers of a tile (4 (n + 1) bytes of SURVEY 8(d)'s count)
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(xv) : "v"(x + tile_of(0) * KR_TILE + 2 * t) : "memory");
    asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(pv) : "v"(rp + tile_of(0) * KR_TILE + 2 * t) : "memory");
    {   // a store in front of the loop, so that every trip sees the same number of younger requests (the first tile's rows: written again below)
        const sk_u4 z4 = {0u, 0u, 0u, 0u};
        asm volatile("global_store_dwordx4 %0, %1, off" : : "v"(y + tile_of(0) * KR_TILE + 2 * t), "v"(z4) : "memory");
    }
    // STATIC register roles (the loop body is two trips long): a register whose load is in flight must not be copied -- the hardware does not
    // interlock a VGPR read with an outstanding load, only s_waitcnt does
    auto trip = [&](int64_t j, sk_u4& xcur, sk_u4& xnext, sk_u2& pcur, sk_u2& pnext, int32_t& rnew_a, int32_t& rnew_b, int32_t& rold_a, int32_t& rold_b) __attribute__((always_inline)) {
        rows_req(tile_of(j + 4), rnew_a, rnew_b);
        if constexpr ((AUX & 4) != 0) asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(pnext) : "v"(rp + tile_of(j + 1) * KR_TILE + 2 * t) : "memory");
        request(quarter(ka, kb), (int)((j + 2) % 3));                          // tile j + 2
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(xnext) : "v"(x + tile_of(j + 1) * KR_TILE + 2 * t) : "memory");
        if constexpr ((AUX & 4) == 0) asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(pnext) : "v"((AUX & 8) ? rp + 2 * t : rp + tile_of(j + 1) * KR_TILE + 2 * t) : "memory");
        // tile j's streams, x and row pointers have landed: younger are the previous trip's store, this trip's 2 + 12 requests, x(j + 1) and its row pointers
        asm volatile("s_waitcnt vmcnt(%0)" : : "n"(SKP_REQ + 5) : "memory");
        asm volatile("" : "+v"(xcur), "+v"(pcur), "+v"(rold_a), "+v"(rold_b)); // (x(j) and the row pointers of tile j + 3 are older still)
        const unsigned char* base = skp_lds + ((size_t)(j % 3) * 4 + w) * SKP_WAVE_BYTES;
        unsigned acc = xcur.x ^ xcur.w ^ pcur.x ^ pcur.y;
#pragma unroll
        for (int u = 0; u < SKP_VAL + SKP_C4; ++u) {
            const sk_u4 v = *reinterpret_cast<const sk_u4*>(base + u * 1024 + 16 * l);
            acc ^= v.x ^ v.y ^ v.z ^ v.w;
        }
#pragma unroll
        for (int u = 0; u < SKP_C1; ++u) acc ^= *reinterpret_cast<const unsigned*>(base + (SKP_VAL + SKP_C4) * 1024 + u * 256 + 4 * l);
        sk_u4 yv; yv.x = acc; yv.y = (unsigned)t; yv.z = 0u; yv.w = 1u;
        const int64_t q = blockIdx.x + j * G;
        // (a tile past the end repeats the last one: the same store again)
        asm volatile("global_store_dwordx4 %0, %1, off nt" : : "v"(y + (q < ntiles ? q : ntiles - 1) * KR_TILE + 2 * t), "v"(yv) : "memory");
        ka = __builtin_amdgcn_readfirstlane(rold_a); kb = __builtin_amdgcn_readfirstlane(rold_b);
    };
#pragma unroll 1
    for (int64_t j = 0; j < trips; j += 2) {                                  // (an odd trip count runs one trip past the end: the last tile again)
        trip(j, xv, xn, pv, pn, a4, b4, a3, b3);
        trip(j + 1, xn, xv, pn, pv, a3, b3, a4, b4);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                           // (requests past the end: nothing of this wave's is in flight when its LDS goes away)
}
}  // namespace kr
extern "C" int32_t kryst_debug_csr_skeleton_persist(kryst_csr_t a, kryst_vec_t x, kryst_vec_t y, int32_t reps, int32_t grid, int32_t aux, double* avg_ms) {
    KR_ARG(a && x && y && avg_ms && reps >= 1 && !a->dist && a->nrows == a->xlen && a->d_row_ptr && a->nrows >= 4 * KR_TILE, "debug_csr_skeleton_persist");
    kryst_ctx_t ctx = a->ctx;
    KR_HIP(hipSetDevice(ctx->device));
    const int64_t nt = ntiles_of(a->nrows);
    KR_ARG(a->nnz <= 7 * a->nrows, "debug_csr_skeleton_persist: sized for at most 7 entries per row");
    const size_t lds = (size_t)3 * 4 * SKP_WAVE_BYTES;
    const unsigned g = (unsigned)std::min<int64_t>(nt, grid > 0 ? grid : ctx->num_cu);
    static bool raised = false;
    if (!raised) {
        KR_HIP(hipFuncSetAttribute((const void*)csr_skeleton_persist_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        KR_HIP(hipFuncSetAttribute((const void*)csr_skeleton_persist_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        KR_HIP(hipFuncSetAttribute((const void*)csr_skeleton_persist_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        KR_HIP(hipFuncSetAttribute((const void*)csr_skeleton_persist_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        raised = true;
    }
    auto once = [&] {
        if (aux == 2) hipLaunchKernelGGL((csr_skeleton_persist_kernel<2>), dim3(g), dim3(KR_T), lds, ctx->s_main, a->d_row_ptr, a->d_col, a->d_val, x->d, y->d, a->nrows, nt, a->nnz);
        else if (aux == 4) hipLaunchKernelGGL((csr_skeleton_persist_kernel<4>), dim3(g), dim3(KR_T), lds, ctx->s_main, a->d_row_ptr, a->d_col, a->d_val, x->d, y->d, a->nrows, nt, a->nnz);
        else if (aux == 8) hipLaunchKernelGGL((csr_skeleton_persist_kernel<8>), dim3(g), dim3(KR_T), lds, ctx->s_main, a->d_row_ptr, a->d_col, a->d_val, x->d, y->d, a->nrows, nt, a->nnz);
        else hipLaunchKernelGGL((csr_skeleton_persist_kernel<0>), dim3(g), dim3(KR_T), lds, ctx->s_main, a->d_row_ptr, a->d_col, a->d_val, x->d, y->d, a->nrows, nt, a->nnz);
    };
    once();
    KR_HIP(hipGetLastError());
    (void)hipEventRecord(ctx->tm0, ctx->s_main);
    for (int r = 0; r < reps; ++r) once();
    (void)hipEventRecord(ctx->tm1, ctx->s_main);
    (void)hipEventSynchronize(ctx->tm1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, ctx->tm0, ctx->tm1);
    *avg_ms = (double)ms / reps;
    KR_HIP(hipGetLastError());
    KR_HIP(hipStreamSynchronize(ctx->s_main));
    return KRYST_OK;
}
