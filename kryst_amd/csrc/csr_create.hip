// Building the device operator: validation (SymbolicSparseRowMat::new_checked, sparse.rs:36-42), the CSR arrays, the lossless
// re-encodings the SpMV kernels stream (CSR-D8 offset codes, CSR-D16 value codes, CSR-P16 row patterns; see spmv.hip), the
// row-partitioned form with its halo plan, and the device-side generator of the synthetic 7-point operators.
#include "csr.h"
#include "ew.h"
#include <algorithm>
#include <unordered_map>

namespace kr {

// host -> device on the context's compute stream, waited for: the compute stream is non-blocking, so nothing issued on the null
// stream orders with it (the round-2 fault)
static int32_t h2d(kryst_ctx_t ctx, void* dst, const void* src, size_t bytes) {
    if (bytes == 0) return KRYST_OK;
    KR_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->s_main));
    KR_HIP(hipStreamSynchronize(ctx->s_main));
    return KRYST_OK;
}

// CSR-P16: number the distinct rows (as sequences of (col - row, value bits)); give up as soon as the tables outgrow LDS
static int32_t upload_patterns(kryst_csr_t a, const std::vector<uint16_t>& pid, const std::vector<uint32_t>& meta /* 2 words per pattern */,
                               const std::vector<int32_t>& poff, const std::vector<double>& pval) {
    kryst_ctx_t ctx = a->ctx;
    KR_HIP(hipMalloc(&a->d_pid, sizeof(uint16_t) * pid.size()));
    KR_HIP(hipMalloc(&a->d_pmeta, sizeof(uint32_t) * 2 * KR_PMAX));
    KR_HIP(hipMalloc(&a->d_poff, sizeof(int32_t) * (KR_TMAX + 8)));
    KR_HIP(hipMalloc(&a->d_pval, sizeof(double) * (KR_TMAX + 8)));
    KR_HIP(hipMemcpyAsync(a->d_pid, pid.data(), sizeof(uint16_t) * pid.size(), hipMemcpyHostToDevice, ctx->s_main));
    KR_HIP(hipMemcpyAsync(a->d_pmeta, meta.data(), sizeof(uint32_t) * meta.size(), hipMemcpyHostToDevice, ctx->s_main));
    if (!poff.empty()) {
        KR_HIP(hipMemcpyAsync(a->d_poff, poff.data(), sizeof(int32_t) * poff.size(), hipMemcpyHostToDevice, ctx->s_main));
        KR_HIP(hipMemcpyAsync(a->d_pval, pval.data(), sizeof(double) * pval.size(), hipMemcpyHostToDevice, ctx->s_main));
    }
    KR_HIP(hipStreamSynchronize(ctx->s_main));
    a->npat = (int32_t)(meta.size() / 2); a->ntab = (int32_t)poff.size();
    // every pattern's base has the columns row-1, row, row+1 at table positions 2, 3, 4 and every row stores its diagonal:
    // the kernel variants that fetch x[row-1 .. row+2] with two gathers and reuse x[row] for the fused dot apply
    a->pat_diag3 = !meta.empty();
    for (size_t p = 0; p + 1 < meta.size() && a->pat_diag3; p += 2) {
        const size_t b = meta[p] & 0xffffu; const int len = (int)(meta[p] >> 16);
        a->pat_diag3 = len == 7 && poff[b + 2] == -1 && poff[b + 3] == 0 && poff[b + 4] == 1 && (meta[p + 1] & 8u);
    }
    // ... and, beyond that, positions 1 and 5 are the columns row - n and row + n with ONE even n <= 1024 in every base (a box of
    // lines of n points, whatever the far couplings at positions 0 and 6 are): spmv_pattern_stage_kernel
    a->pat_stage_n = 0;
    if (a->pat_diag3) {
        int32_t nn = -1;
        for (size_t p = 0; p + 1 < meta.size() && nn != 0; p += 2) {
            const size_t b = meta[p] & 0xffffu;
            const int32_t lo = poff[b + 1], hi = poff[b + 5];
            if (lo >= 0 || hi != -lo || (nn > 0 && hi != nn)) nn = 0; else nn = hi;
        }
        if (nn >= 8 && nn <= 1024 && nn % 2 == 0) a->pat_stage_n = nn;
        a->pat_far_uniform = a->pat_stage_n > 0;
        for (size_t p = 0; p + 1 < meta.size() && a->pat_far_uniform; p += 2) {
            const size_t b = meta[p] & 0xffffu;
            if (p == 0) { a->pat_far_lo = poff[b]; a->pat_far_hi = poff[b + 6]; }
            else a->pat_far_uniform = poff[b] == a->pat_far_lo && poff[b + 6] == a->pat_far_hi;
        }
    }
    return KRYST_OK;
}
static int32_t build_patterns(kryst_csr_t a, const std::vector<int32_t>& rp, const std::vector<int32_t>& col, const double* val) {
    const int64_t n = a->nrows;
    if (n == 0 || a->nnz == 0) return KRYST_OK;
    struct Key { uint64_t h; int32_t row; };
    std::unordered_multimap<uint64_t, int> by_hash;               // hash -> pattern id
    std::vector<uint32_t> meta; std::vector<int32_t> poff; std::vector<double> pval;
    std::vector<uint16_t> pid((size_t)((n + KR_TILE - 1) / KR_TILE * KR_TILE + KR_TILE), 0);
    auto same = [&](int p, int64_t i) {
        const int b = (int)(meta[p] & 0xffffu), len = (int)(meta[p] >> 16);
        if (len != rp[i + 1] - rp[i]) return false;
        for (int e = 0; e < len; ++e) {
            const int32_t k = rp[i] + e;
            if (poff[b + e] != col[k] - (int32_t)i || memcmp(&pval[b + e], &val[k], 8) != 0) return false;
        }
        return true;
    };
    int last = -1, maxlen = 0;
    for (int64_t i = 0; i < n; ++i) {
        const int len = rp[i + 1] - rp[i];
        if (len > 0xffff) return KRYST_OK;
        if (last >= 0 && same(last, i)) { pid[i] = (uint16_t)last; continue; }     // neighbouring rows usually repeat
        uint64_t h = 1469598103934665603ull ^ (uint64_t)len;
        for (int32_t k = rp[i]; k < rp[i + 1]; ++k) {
            uint64_t bits; memcpy(&bits, &val[k], 8);
            h = (h ^ (uint64_t)(uint32_t)(col[k] - (int32_t)i)) * 1099511628211ull;
            h = (h ^ bits) * 1099511628211ull;
        }
        int found = -1;
        auto range = by_hash.equal_range(h);
        for (auto it = range.first; it != range.second; ++it) if (same(it->second, i)) { found = it->second; break; }
        if (found < 0) {
            if ((int)meta.size() == KR_PMAX || (int)poff.size() + len > KR_TMAX) return KRYST_OK;     // not a pattern matrix
            found = (int)meta.size();
            meta.push_back((uint32_t)poff.size() | ((uint32_t)len << 16));
            for (int32_t k = rp[i]; k < rp[i + 1]; ++k) { poff.push_back(col[k] - (int32_t)i); pval.push_back(val[k]); }
            by_hash.emplace(h, found);
            maxlen = std::max(maxlen, len);
        }
        pid[i] = (uint16_t)found; last = found;
    }
    // Express every pattern as (base, presence mask): a pattern of at most 16 entries that is a sub-sequence of a longer
    // pattern's (offset, value) sequence shares that pattern's table entries (boundary rows of a stencil are the interior
    // row with entries removed).  Longest patterns first; a base is padded to a multiple of the kernel's unroll factor
    // (offset 0, value 0: never used, the kernel predicates on length and mask; the padding only keeps the clamp-free
    // immediate-offset reads in bounds).
    const int U = maxlen <= 7 ? 7 : 8;
    {
        const size_t np = meta.size();
        std::vector<int> order(np);
        for (size_t p = 0; p < np; ++p) order[p] = (int)p;
        std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return (meta[x] >> 16) > (meta[y] >> 16); });
        std::vector<uint32_t> m2(2 * np, 0u); std::vector<int32_t> o2; std::vector<double> v2;
        struct Base { uint32_t start; int len; };
        std::vector<Base> bases;
        for (int p : order) {
            const int b = (int)(meta[p] & 0xffffu), len = (int)(meta[p] >> 16);
            bool placed = false;
            if (len <= 16)
                for (const Base& B : bases) {
                    if (B.len > 16 || B.len < len) continue;
                    uint32_t mask = 0; int e = 0;
                    for (int f = 0; f < B.len && e < len; ++f)
                        if (o2[B.start + f] == poff[b + e] && memcmp(&v2[B.start + f], &pval[b + e], 8) == 0) { mask |= 1u << f; ++e; }
                    if (e == len) { m2[2 * p] = B.start | ((uint32_t)B.len << 16); m2[2 * p + 1] = mask; placed = true; break; }
                }
            if (placed) continue;
            const uint32_t start = (uint32_t)o2.size();
            for (int e = 0; e < len; ++e) { o2.push_back(poff[b + e]); v2.push_back(pval[b + e]); }
            while (o2.size() % (size_t)U) { o2.push_back(0); v2.push_back(0.0); }
            bases.push_back(Base{start, len});
            m2[2 * p] = start | ((uint32_t)len << 16);
            m2[2 * p + 1] = len >= 16 ? 0xffffu : ((1u << len) - 1u);
        }
        if ((int)o2.size() > KR_TMAX || o2.size() > 0xffffu) return KRYST_OK;
        meta.swap(m2); poff.swap(o2); pval.swap(v2);
    }
    a->pat_unroll = U; a->pat_single = maxlen <= U;
    return upload_patterns(a, pid, meta, poff, pval);
}


// ---------------------------------------------------------------- CSR-DIA (built on the device from the CSR-D8 form)
// used[] : 256-bit map of the offset codes that occur
__global__ __launch_bounds__(256) void dia_usage_kernel(const uint8_t* __restrict__ code, int64_t nnz, unsigned* used) {
    __shared__ unsigned lds[8];
    if (threadIdx.x < 8) lds[threadIdx.x] = 0u;
    __syncthreads();
    unsigned mine[8] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < nnz; k += (int64_t)gridDim.x * blockDim.x) {
        const unsigned c = code[k];
#pragma unroll
        for (int w = 0; w < 8; ++w) mine[w] |= ((c >> 5) == (unsigned)w) ? (1u << (c & 31u)) : 0u;
    }
#pragma unroll
    for (int w = 0; w < 8; ++w) if (mine[w]) atomicOr(&lds[w], mine[w]);
    __syncthreads();
    if (threadIdx.x < 8 && lds[threadIdx.x]) atomicOr(&used[threadIdx.x], lds[threadIdx.x]);
}
// prec[a] bit b: in some row an entry of diagonal a is stored directly before one of diagonal b;  flags[0] |= 1 when a row holds a
// diagonal twice or is longer than the diagonal count (not a DIA operator)
__global__ __launch_bounds__(256) void dia_precede_kernel(const int32_t* __restrict__ rp, const uint8_t* __restrict__ code, int64_t n,
                                                          const uint8_t* __restrict__ idx_of_code, int nd, unsigned* prec, unsigned* flags) {
    __shared__ unsigned lds[KR_DIA_MAX];
    if (threadIdx.x < KR_DIA_MAX) lds[threadIdx.x] = 0u;
    __syncthreads();
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        unsigned seen = 0u; int prev = -1;
        for (int32_t k = rp[i]; k < rp[i + 1]; ++k) {
            const int d = idx_of_code[code[k]];
            if ((seen >> d) & 1u) atomicOr(&flags[0], 1u);
            seen |= 1u << d;
            if (prev >= 0 && !((lds[prev] >> d) & 1u)) atomicOr(&lds[prev], 1u << d);
            prev = d;
        }
        if (rp[i + 1] - rp[i] > nd) atomicOr(&flags[0], 1u);
    }
    __syncthreads();
    if (threadIdx.x < KR_DIA_MAX && lds[threadIdx.x]) atomicOr(&prec[threadIdx.x], lds[threadIdx.x]);
}
__global__ __launch_bounds__(256) void dia_fill_kernel(double* dia, int64_t count) {
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < count; k += (int64_t)gridDim.x * blockDim.x)
        dia[k] = __longlong_as_double((long long)KR_DIA_ABSENT);
}
// one thread per row: its entries go to their diagonals' streams (rank_of_code: position of the code's diagonal in the stored order)
__global__ __launch_bounds__(256) void dia_scatter_kernel(const int32_t* __restrict__ rp, const uint8_t* __restrict__ code, const double* __restrict__ val,
                                                          int64_t n, const uint8_t* __restrict__ rank_of_code, double* dia, int64_t stride, unsigned* flags) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    for (int32_t k = rp[i]; k < rp[i + 1]; ++k) {
        const double v = val[k];
        if ((unsigned long long)__double_as_longlong(v) == KR_DIA_ABSENT) atomicOr(&flags[0], 2u);     // a stored value with the marker's bits
        dia[(int64_t)rank_of_code[code[k]] * stride + i] = v;
    }
}

// Build the CSR-DIA streams when the operator qualifies (see csr.h); not an error when it does not.  KRYST_SPMV_DIA: 0 never,
// 1 (default) when the operator has no CSR-D16 / CSR-P16 form (those stream fewer bytes), 2 whenever it qualifies.
static int32_t build_dia(kryst_csr_t a) {
    kryst_ctx_t ctx = a->ctx;
    const int want = env_int("KRYST_SPMV_DIA", 1);
    const int64_t n = a->nrows, nnz = a->nnz;
    if (want == 0 || !a->d_code || !a->d_dict || n == 0 || nnz == 0) return KRYST_OK;
    if (want < 2 && a->d_code16) return KRYST_OK;
    struct Scratch { unsigned* u = nullptr; uint8_t* b = nullptr; ~Scratch() { (void)hipFree(u); (void)hipFree(b); } } sc;
    KR_HIP(hipMalloc(&sc.u, sizeof(unsigned) * (16 + KR_DIA_MAX))); KR_HIP(hipMalloc(&sc.b, 512));
    KR_HIP(hipMemsetAsync(sc.u, 0, sizeof(unsigned) * (16 + KR_DIA_MAX), ctx->s_main));
    unsigned* d_used = sc.u; unsigned* d_prec = sc.u + 8; unsigned* d_flags = sc.u + 8 + KR_DIA_MAX;
    const unsigned eg = (unsigned)std::min<int64_t>(4096, (nnz + 255) / 256), rg = (unsigned)((n + 255) / 256);
    hipLaunchKernelGGL(dia_usage_kernel, dim3(eg), dim3(256), 0, ctx->s_main, a->d_code, nnz, d_used);
    KR_HIP(hipGetLastError());
    unsigned used[8]; int32_t dict[256];
    KR_HIP(hipMemcpyAsync(used, d_used, sizeof used, hipMemcpyDeviceToHost, ctx->s_main));
    KR_HIP(hipMemcpyAsync(dict, a->d_dict, sizeof dict, hipMemcpyDeviceToHost, ctx->s_main));
    KR_HIP(hipStreamSynchronize(ctx->s_main));
    std::vector<int> codes;
    for (int c = 0; c < 256; ++c) if ((used[c >> 5] >> (c & 31)) & 1u) codes.push_back(c);
    const int nd = (int)codes.size();
    if (nd == 0 || nd > KR_DIA_MAX) return KRYST_OK;
    if ((double)nd * (double)n > 1.125 * (double)nnz + 64.0) return KRYST_OK;       // diagonals too sparsely filled: 8 nd n + 16 n would exceed CSR-D8's bytes
    // stored order of the diagonals: ascending offset for a plain operator (columns ascend inside a row); with halo columns the
    // local numbering breaks that, so the order is read off the rows (an entry of diagonal a directly before one of diagonal b)
    std::sort(codes.begin(), codes.end(), [&](int x, int y) { return dict[x] < dict[y]; });
    uint8_t idx_of_code[256] = {0}, rank_of_code[256] = {0};
    for (int d = 0; d < nd; ++d) idx_of_code[codes[d]] = (uint8_t)d;
    KR_HIP(hipMemcpyAsync(sc.b, idx_of_code, 256, hipMemcpyHostToDevice, ctx->s_main));
    hipLaunchKernelGGL(dia_precede_kernel, dim3(rg), dim3(256), 0, ctx->s_main, a->d_row_ptr, a->d_code, n, sc.b, nd, d_prec, d_flags);
    KR_HIP(hipGetLastError());
    unsigned prec[KR_DIA_MAX], flags[1];
    KR_HIP(hipMemcpyAsync(prec, d_prec, sizeof prec, hipMemcpyDeviceToHost, ctx->s_main));
    KR_HIP(hipMemcpyAsync(flags, d_flags, sizeof flags, hipMemcpyDeviceToHost, ctx->s_main));
    KR_HIP(hipStreamSynchronize(ctx->s_main));
    if (flags[0]) return KRYST_OK;
    std::vector<int> order;                                   // topological order of the "directly before" relation (Kahn; ties: ascending offset)
    {
        int indeg[KR_DIA_MAX] = {0};
        for (int x = 0; x < nd; ++x) for (int y = 0; y < nd; ++y) if ((prec[x] >> y) & 1u) ++indeg[y];
        std::vector<bool> done((size_t)nd, false);
        for (int step = 0; step < nd; ++step) {
            int pick = -1;
            for (int x = 0; x < nd && pick < 0; ++x) if (!done[x] && indeg[x] == 0) pick = x;
            if (pick < 0) return KRYST_OK;                    // a cycle: rows disagree about the order of two diagonals
            done[pick] = true; order.push_back(pick);
            for (int y = 0; y < nd; ++y) if ((prec[pick] >> y) & 1u) --indeg[y];
        }
    }
    // a row's entries need not be ADJACENT diagonals, so "directly before" alone does not order every pair a row holds; the closure
    // does: require the order to be the unique linear extension on every pair that some row relates (checked by the scatter's
    // companion below: each row's diagonal ranks must ascend)
    for (int r = 0; r < nd; ++r) rank_of_code[codes[order[r]]] = (uint8_t)r;
    const int64_t stride = (n + KR_TILE - 1) / KR_TILE * KR_TILE + KR_TILE;
    if (hipMalloc(&a->d_dia, sizeof(double) * (size_t)stride * nd) != hipSuccess) { (void)hipGetLastError(); a->d_dia = nullptr; return KRYST_OK; }   // optional form
    KR_HIP(hipMemcpyAsync(sc.b + 256, rank_of_code, 256, hipMemcpyHostToDevice, ctx->s_main));
    KR_HIP(hipMemsetAsync(d_flags, 0, sizeof(unsigned) * 8, ctx->s_main));
    hipLaunchKernelGGL(dia_fill_kernel, dim3(4096), dim3(256), 0, ctx->s_main, a->d_dia, stride * nd);
    hipLaunchKernelGGL(dia_scatter_kernel, dim3(rg), dim3(256), 0, ctx->s_main, a->d_row_ptr, a->d_code, a->d_val, n, sc.b + 256, a->d_dia, stride, d_flags);
    // ranks ascend inside every row?  (reuses the precedence kernel with the ranks as indices: bit b of prec[a] with b <= a is a violation)
    KR_HIP(hipMemsetAsync(d_prec, 0, sizeof(unsigned) * KR_DIA_MAX, ctx->s_main));
    hipLaunchKernelGGL(dia_precede_kernel, dim3(rg), dim3(256), 0, ctx->s_main, a->d_row_ptr, a->d_code, n, sc.b + 256, nd, d_prec, d_flags);
    KR_HIP(hipGetLastError());
    KR_HIP(hipMemcpyAsync(prec, d_prec, sizeof prec, hipMemcpyDeviceToHost, ctx->s_main));
    KR_HIP(hipMemcpyAsync(flags, d_flags, sizeof flags, hipMemcpyDeviceToHost, ctx->s_main));
    KR_HIP(hipStreamSynchronize(ctx->s_main));
    bool ok = flags[0] == 0;
    for (int x = 0; x < nd && ok; ++x) if (prec[x] & ((2u << x) - 1u)) ok = false;
    if (!ok) { (void)hipFree(a->d_dia); a->d_dia = nullptr; return KRYST_OK; }
    a->dia_stride = stride; a->dia_nd = nd;
    a->dia_min = 0; a->dia_max = 0;
    for (int r = 0; r < nd; ++r) {
        const int32_t off = dict[codes[order[r]]];
        a->dia_off[r] = off;
        a->dia_min = std::min(a->dia_min, off); a->dia_max = std::max(a->dia_max, off);
    }
    return KRYST_OK;
}

// ---------------------------------------------------------------- tile order (plane-structured operators)
// An operator whose entries sit near the main diagonal and near ONE far pair of diagonals +-b (every stencil on a structured grid:
// b = one grid plane) re-reads x[row - b] and x[row + b] one and two planes after x[row] went by: in natural tile order an XCD's L2
// has to hold 16 b bytes of x for that (4 MiB at 512^3 -- it does not), and because consecutive tiles go to different XCDs every
// line of x is fetched by two or three of them.  The tiles are therefore handed out in SLAB order: the plane is cut into S = 8, 16, ...
// segments of L rows (16 L bytes <= 1 MiB), XCD x walks segment x of plane 0, 1, 2, ... (then segment x + 7, ...), so that an XCD's
// window of x is 2 L rows and only the segments' edges are shared.  Which rows a tile holds, the order inside a row and the order in
// which the tile partials of the fused inner products are folded do not change: same bits.  order1 / order8: slot -> tile (or -1)
// in the kernels' slot numbering for runs of 1 / 8 consecutive slots per XCD (spmv.hip: tile -> workgroup map).
static int32_t build_tile_order(kryst_csr_t a) {
    kryst_ctx_t ctx = a->ctx;
    if (env_int("KRYST_SPMV_ORDER", 1) == 0 || !a->d_code || !a->d_dict || a->nnz == 0 || a->dist || a->ntiles < 64) return KRYST_OK;
    struct Scratch { unsigned* u = nullptr; ~Scratch() { (void)hipFree(u); } } sc;
    KR_HIP(hipMalloc(&sc.u, sizeof(unsigned) * 8));
    KR_HIP(hipMemsetAsync(sc.u, 0, sizeof(unsigned) * 8, ctx->s_main));
    hipLaunchKernelGGL(dia_usage_kernel, dim3((unsigned)std::min<int64_t>(4096, (a->nnz + 255) / 256)), dim3(256), 0, ctx->s_main, a->d_code, a->nnz, sc.u);
    KR_HIP(hipGetLastError());
    unsigned used[8]; int32_t dict[256];
    KR_HIP(hipMemcpyAsync(used, sc.u, sizeof used, hipMemcpyDeviceToHost, ctx->s_main));
    KR_HIP(hipMemcpyAsync(dict, a->d_dict, sizeof dict, hipMemcpyDeviceToHost, ctx->s_main));
    KR_HIP(hipStreamSynchronize(ctx->s_main));
    int64_t b = 0;
    for (int c = 0; c < 256; ++c) if ((used[c >> 5] >> (c & 31)) & 1u) b = std::max<int64_t>(b, std::llabs((long long)dict[c]));
    if (b < env_int("KRYST_SPMV_ORDER_MIN_PLANE", 65536)) return KRYST_OK;          // 16 b bytes < 1 MiB: the window fits as it is
    int64_t w = 0;                                                                      // how far an entry strays from 0 / +-b
    for (int c = 0; c < 256; ++c) if ((used[c >> 5] >> (c & 31)) & 1u) {
        const int64_t o = std::llabs((long long)dict[c]);
        w = std::max(w, std::min(o, b - o));
    }
    const int64_t lmax = env_int("KRYST_SPMV_ORDER_SEG_ROWS", 65536);
    const int64_t S = 8 * ((b + 8 * lmax - 1) / (8 * lmax));
    const int64_t L = b / S;                                                            // (a tile belongs to the segment its first row falls into)
    if (L < 4 * KR_TILE || 8 * w > L) return KRYST_OK;                                  // segments of a few tiles / mostly edge: not worth it
    std::vector<int32_t> seq[8];
    for (int64_t p = 0; p < S / 8; ++p)
        for (int64_t q = 0; q < a->ntiles; ++q) {
            const int64_t seg = std::min((q * KR_TILE) % b * S / b, S - 1);
            if (seg / 8 == p) seq[(seg + p) % 8].push_back((int32_t)q);              // (+ p: segments of 4 and 5 tiles alternate; mix them)
        }
    size_t longest = 0;
    for (int x = 0; x < 8; ++x) longest = std::max(longest, seq[x].size());
    const size_t c1 = longest, c8 = (longest + 7) / 8 * 8;
    std::vector<int32_t> order(8 * c1 + 8 * c8, -1);
    for (int x = 0; x < 8; ++x)
        for (size_t li = 0; li < seq[x].size(); ++li) {
            order[li * 8 + x] = seq[x][li];
            order[8 * c1 + ((li / 8) * 8 + x) * 8 + li % 8] = seq[x][li];
        }
    if (hipMalloc(&a->d_tile_order, sizeof(int32_t) * order.size()) != hipSuccess) { (void)hipGetLastError(); a->d_tile_order = nullptr; return KRYST_OK; }   // optional
    KR_TRY(h2d(ctx, a->d_tile_order, order.data(), sizeof(int32_t) * order.size()));
    a->order_slots1 = (int64_t)(8 * c1); a->order_slots8 = (int64_t)(8 * c8); a->order_plane = b;
    return KRYST_OK;
}

// ---------------------------------------------------------------- creation
static int32_t upload_csr(kryst_csr_t a, const std::vector<int32_t>& rp, const std::vector<int32_t>& col, const double* val) {
    kryst_ctx_t ctx = a->ctx;
    const size_t nnz = (size_t)a->nnz;
    KR_HIP(hipMalloc(&a->d_row_ptr, sizeof(int32_t) * (rp.size() + 8)));
    KR_HIP(hipMalloc(&a->d_col, sizeof(int32_t) * (nnz + 8)));
    KR_HIP(hipMalloc(&a->d_val, sizeof(double) * (nnz + 8)));
    KR_HIP(hipMemsetAsync(a->d_col + nnz, 0, sizeof(int32_t) * 8, ctx->s_main));
    KR_HIP(hipMemsetAsync(a->d_val + nnz, 0, sizeof(double) * 8, ctx->s_main));
    KR_HIP(hipMemcpyAsync(a->d_row_ptr, rp.data(), sizeof(int32_t) * rp.size(), hipMemcpyHostToDevice, ctx->s_main));
    if (nnz) {
        KR_HIP(hipMemcpyAsync(a->d_col, col.data(), sizeof(int32_t) * nnz, hipMemcpyHostToDevice, ctx->s_main));
        KR_HIP(hipMemcpyAsync(a->d_val, val, sizeof(double) * nnz, hipMemcpyHostToDevice, ctx->s_main));
    }
    KR_HIP(hipStreamSynchronize(ctx->s_main));
    a->ntiles = ntiles_of(a->nrows);
    {   // CSR-D8: one byte per entry when the operator has <= 256 distinct (col - row) offsets
        std::vector<int32_t> dict; dict.reserve(256);
        std::unordered_map<int32_t, int> code_of;
        std::vector<uint8_t> codes(nnz + 32, 0);
        bool ok = nnz > 0;
        int32_t last_d = 0; int last_code = -1;                    // consecutive entries often repeat an offset
        for (int64_t i = 0; i < a->nrows && ok; ++i)
            for (int32_t k = rp[i]; k < rp[i + 1]; ++k) {
                const int32_t d = col[k] - (int32_t)i;
                int code;
                if (last_code >= 0 && d == last_d) code = last_code;
                else {
                    auto it = code_of.find(d);
                    if (it != code_of.end()) code = it->second;
                    else {
                        if (dict.size() == 256) { ok = false; break; }
                        dict.push_back(d); code = (int)dict.size() - 1; code_of.emplace(d, code);
                    }
                    last_d = d; last_code = code;
                }
                codes[k] = (uint8_t)code;
            }
        if (ok) {
            dict.resize(256, 0);
            KR_HIP(hipMalloc(&a->d_code, codes.size()));
            KR_HIP(hipMalloc(&a->d_dict, sizeof(int32_t) * 256));
            KR_HIP(hipMemcpyAsync(a->d_code, codes.data(), codes.size(), hipMemcpyHostToDevice, ctx->s_main));
            KR_HIP(hipMemcpyAsync(a->d_dict, dict.data(), sizeof(int32_t) * 256, hipMemcpyHostToDevice, ctx->s_main));
            KR_HIP(hipStreamSynchronize(ctx->s_main));
            // CSR-D16: additionally <= 256 distinct value bit patterns -> one 16-bit word per entry
            std::vector<double> vdict; vdict.reserve(256);
            std::unordered_map<uint64_t, int> vcode_of;
            std::vector<uint16_t> c16(nnz + 32, 0);
            bool vok = true;
            uint64_t last_b = 0; int last_vc = -1;
            for (size_t k = 0; k < nnz && vok; ++k) {
                uint64_t bits; memcpy(&bits, &val[k], 8);
                int vc;
                if (last_vc >= 0 && bits == last_b) vc = last_vc;
                else {
                    auto it = vcode_of.find(bits);
                    if (it != vcode_of.end()) vc = it->second;
                    else {
                        if (vdict.size() == 256) { vok = false; break; }
                        vdict.push_back(val[k]); vc = (int)vdict.size() - 1; vcode_of.emplace(bits, vc);
                    }
                    last_b = bits; last_vc = vc;
                }
                c16[k] = (uint16_t)((vc << 8) | codes[k]);
            }
            if (vok) {
                vdict.resize(256, 0.0);
                KR_HIP(hipMalloc(&a->d_code16, sizeof(uint16_t) * c16.size()));
                KR_HIP(hipMalloc(&a->d_vdict, sizeof(double) * 256));
                KR_HIP(hipMemcpyAsync(a->d_code16, c16.data(), sizeof(uint16_t) * c16.size(), hipMemcpyHostToDevice, ctx->s_main));
                KR_HIP(hipMemcpyAsync(a->d_vdict, vdict.data(), sizeof(double) * 256, hipMemcpyHostToDevice, ctx->s_main));
                KR_HIP(hipStreamSynchronize(ctx->s_main));
            }
        }
    }
    const double per_slice = a->nrows > 0 ? (double)a->nnz / (double)((a->nrows + 127) / 128) : 0.0;
    a->slots = per_slice <= 256.0 ? 2 : (per_slice <= 512.0 ? 4 : 7);
    KR_TRY(build_patterns(a, rp, col, val));
    KR_TRY(build_dia(a));
    KR_TRY(build_tile_order(a));
    return KRYST_OK;
}

// SymbolicSparseRowMat::new_checked (sparse.rs:36-42): monotone row_ptr, in-bounds sorted duplicate-free columns
template <class P, class I>
static int32_t check_csr(int64_t nrows, int64_t ncols, const P* rp, const I* col) {
    if (rp[0] != 0) { set_error("row_ptr[0] != 0"); return KRYST_ERR_CSR; }
    for (int64_t i = 0; i < nrows; ++i) {
        if (rp[i + 1] < rp[i]) { set_error("row_ptr not monotone at row %lld", (long long)i); return KRYST_ERR_CSR; }
        for (int64_t k = (int64_t)rp[i]; k < (int64_t)rp[i + 1]; ++k) {
            if ((int64_t)col[k] < 0 || (int64_t)col[k] >= ncols) { set_error("column out of range in row %lld", (long long)i); return KRYST_ERR_CSR; }
            if (k > (int64_t)rp[i] && col[k] <= col[k - 1]) { set_error("columns not strictly ascending in row %lld", (long long)i); return KRYST_ERR_CSR; }
        }
    }
    return KRYST_OK;
}

template <class P, class I>
static int32_t create_local(kryst_ctx_t ctx, int64_t nrows, int64_t ncols, const P* rp, const I* col, const double* val,
                            kryst_csr_t* out) {
    KR_ARG(ctx && rp && out && nrows >= 0 && ncols >= 0, "csr_create");
    KR_ARG(nrows < (1ll << 31) - KR_TILE && ncols < (1ll << 31), "dimension exceeds int32 device indexing");
    const int64_t nnz = (int64_t)rp[nrows];
    KR_ARG(nnz < (1ll << 31) - 16, "nnz exceeds int32 device indexing");
    KR_ARG(nnz == 0 || (col && val), "csr_create: null arrays");
    KR_TRY(check_csr(nrows, ncols, rp, col));
    KR_HIP(hipSetDevice(ctx->device));
    kryst_csr_t a = new kryst_csr_s();
    a->ctx = ctx; a->nrows = nrows; a->ncols = ncols; a->xlen = ncols; a->nnz = nnz;
    std::vector<int32_t> rp32((size_t)nrows + 1), c32((size_t)nnz);
    for (int64_t i = 0; i <= nrows; ++i) rp32[i] = (int32_t)rp[i];
    for (int64_t k = 0; k < nnz; ++k) c32[k] = (int32_t)col[k];
    int32_t rc = upload_csr(a, rp32, c32, val);
    if (rc == KRYST_OK) rc = csr_place(a);
    if (rc != KRYST_OK) { kryst_csr_destroy(a); return rc; }
    *out = a;
    return KRYST_OK;
}

}  // namespace kr

using namespace kr;

extern "C" {

int32_t kryst_csr_create(kryst_ctx_t ctx, int64_t nrows, int64_t ncols, const uint64_t* row_ptr, const uint64_t* col_idx,
                         const double* vals, kryst_csr_t* out) {
    KR_ARG(ctx && ctx->nranks == 1, "csr_create needs a single-rank context (use kryst_csr_create_dist)");
    return create_local(ctx, nrows, ncols, (const int64_t*)row_ptr, (const int64_t*)col_idx, vals, out);
}

int32_t kryst_csr_create_i32(kryst_ctx_t ctx, int64_t nrows, int64_t ncols, const int64_t* row_ptr, const int32_t* col_idx,
                             const double* vals, kryst_csr_t* out) {
    KR_ARG(ctx && ctx->nranks == 1, "csr_create needs a single-rank context (use kryst_csr_create_dist)");
    return create_local(ctx, nrows, ncols, row_ptr, col_idx, vals, out);
}

// Phase 1 of kryst_csr_create_dist: everything that involves this rank alone (validation, halo receive plan, local column
// numbering, upload).  No collective is entered here, so a rank may fail without the others noticing -- the caller agrees on the
// outcome across ranks before phase 2.
static int32_t create_dist_local(kryst_csr_t a, int64_t n_global, const int64_t* row_offsets, const int64_t* row_ptr,
                                 const int64_t* col_global, const double* vals) {
    kryst_ctx_t ctx = a->ctx;
    const int P = ctx->nranks, me = ctx->rank;
    KR_ARG(row_offsets[0] == 0 && row_offsets[P] == n_global, "row_offsets must cover [0, n_global)");
    const int64_t lo = row_offsets[me], hi = row_offsets[me + 1], nloc = hi - lo;
    KR_ARG(nloc >= 0, "row_offsets must be non-decreasing");
    const int64_t nnz = row_ptr[nloc];
    KR_ARG(nloc < (1ll << 31) - KR_TILE && nnz < (1ll << 31) - 16, "local block exceeds int32 device indexing");
    KR_TRY(check_csr(nloc, n_global, row_ptr, col_global));
    a->nrows = nloc; a->ncols = n_global; a->xlen = nloc; a->nnz = nnz; a->dist = true;
    a->row_offsets.assign(row_offsets, row_offsets + P + 1);
    HaloPlan& pl = a->plan;
    halo_recv_plan(me, P, row_offsets, nloc, row_ptr, col_global, &pl);
    KR_ARG(nloc + pl.total_recv < (1ll << 31), "local + halo columns exceed int32");
    // local column numbering: owned -> c - lo, halo -> nloc + slot; classify rows / tiles
    std::vector<int32_t> rp32((size_t)nloc + 1), c32((size_t)nnz);
    const int64_t ntiles = ntiles_of(nloc);
    std::vector<char> tile_bnd((size_t)ntiles, 0);
    for (int64_t i = 0; i <= nloc; ++i) rp32[i] = (int32_t)row_ptr[i];
    for (int64_t i = 0; i < nloc; ++i)
        for (int64_t k = row_ptr[i]; k < row_ptr[i + 1]; ++k) {
            const int64_t c = col_global[k];
            if (c >= lo && c < hi) c32[k] = (int32_t)(c - lo);
            else { c32[k] = (int32_t)(nloc + halo_slot(pl, row_offsets, c)); tile_bnd[i / KR_TILE] = 1; }
        }
    KR_TRY(upload_csr(a, rp32, c32, vals));
    std::vector<int32_t> ti, tb;
    for (int64_t q = 0; q < ntiles; ++q) (tile_bnd[q] ? tb : ti).push_back((int32_t)q);
    a->n_interior = (int64_t)ti.size(); a->n_boundary = (int64_t)tb.size();
    a->interior_first = (!ti.empty() && (int64_t)ti.back() - (int64_t)ti.front() + 1 == (int64_t)ti.size()) ? (int64_t)ti.front() : -1;
    KR_HIP(hipMalloc(&a->d_tiles_interior, sizeof(int32_t) * (ti.size() + 1)));
    KR_HIP(hipMalloc(&a->d_tiles_boundary, sizeof(int32_t) * (tb.size() + 1)));
    KR_TRY(h2d(ctx, a->d_tiles_interior, ti.data(), sizeof(int32_t) * ti.size()));
    KR_TRY(h2d(ctx, a->d_tiles_boundary, tb.data(), sizeof(int32_t) * tb.size()));
    KR_HIP(hipMalloc(&pl.d_halo, sizeof(double) * (size_t)(pl.total_recv + 2)));
    KR_HIP(hipMemsetAsync(pl.d_halo, 0, sizeof(double) * (size_t)(pl.total_recv + 2), ctx->s_main));
    KR_HIP(hipStreamSynchronize(ctx->s_main));
    return KRYST_OK;
}

// device scratch of the index-list exchange, freed on every path
struct DistScratch {
    int64_t *cnt_s = nullptr, *cnt_r = nullptr, *req = nullptr, *ans = nullptr;
    ~DistScratch() { (void)hipFree(cnt_s); (void)hipFree(cnt_r); (void)hipFree(req); (void)hipFree(ans); }
};

// Phase 2: every owner learns which of its rows the others need (one all-gather of counts, one exchange of index lists).
// All ranks enter it together (the caller has agreed that phase 1 succeeded everywhere).
static int32_t create_dist_exchange(kryst_csr_t a) {
    kryst_ctx_t ctx = a->ctx;
    const int P = ctx->nranks, me = ctx->rank;
    HaloPlan& pl = a->plan;
    const int64_t lo = a->row_offsets[me], hi = a->row_offsets[me + 1];
    pl.send_counts.assign(P, 0); pl.send_off.assign(P, 0);
    if (P == 1) return KRYST_OK;
    DistScratch d;
    KR_HIP(hipMalloc(&d.cnt_s, sizeof(int64_t) * P));
    KR_HIP(hipMalloc(&d.cnt_r, sizeof(int64_t) * (size_t)P * P));
    KR_TRY(h2d(ctx, d.cnt_s, pl.recv_counts.data(), sizeof(int64_t) * P));
    KR_TRY(comm_all_gather_i64(ctx, d.cnt_s, d.cnt_r, P, ctx->s_main));
    std::vector<int64_t> cnt((size_t)P * P);
    KR_HIP(hipMemcpyAsync(cnt.data(), d.cnt_r, sizeof(int64_t) * cnt.size(), hipMemcpyDeviceToHost, ctx->s_main));
    KR_HIP(hipStreamSynchronize(ctx->s_main));
    for (int p = 0; p < P; ++p) pl.send_counts[p] = cnt[(size_t)p * P + me];   // what p receives from me
    pl.total_send = 0;
    for (int p = 0; p < P; ++p) { pl.send_off[p] = pl.total_send; pl.total_send += pl.send_counts[p]; }
    KR_HIP(hipMalloc(&d.req, sizeof(int64_t) * (size_t)(pl.total_recv + 1)));
    KR_HIP(hipMalloc(&d.ans, sizeof(int64_t) * (size_t)(pl.total_send + 1)));
    KR_TRY(h2d(ctx, d.req, pl.recv_cols.data(), sizeof(int64_t) * (size_t)pl.total_recv));
    KR_TRY(comm_exchange(ctx, d.req, pl.recv_counts.data(), pl.recv_off.data(), d.ans, pl.send_counts.data(),
                         pl.send_off.data(), false, ctx->s_main));
    std::vector<int64_t> ans((size_t)pl.total_send);
    if (pl.total_send) KR_HIP(hipMemcpyAsync(ans.data(), d.ans, sizeof(int64_t) * pl.total_send, hipMemcpyDeviceToHost, ctx->s_main));
    KR_HIP(hipStreamSynchronize(ctx->s_main));
    std::vector<int32_t> sidx((size_t)pl.total_send);
    bool contig = true;
    for (int p = 0; p < P; ++p)
        for (int64_t k = 0; k < pl.send_counts[p]; ++k) {
            const int64_t g = ans[pl.send_off[p] + k];
            KR_ARG(g >= lo && g < hi, "halo request for a row this rank does not own");
            sidx[pl.send_off[p] + k] = (int32_t)(g - lo);
            if (k > 0 && g != ans[pl.send_off[p] + k - 1] + 1) contig = false;
        }
    a->send_contiguous = contig;
    if (contig)      // send_off then holds the first local row of each run (used as the offset into x)
        for (int p = 0; p < P; ++p) if (pl.send_counts[p]) pl.send_off[p] = sidx[pl.send_off[p]];
    KR_HIP(hipMalloc(&pl.d_send_idx, sizeof(int32_t) * (sidx.size() + 1)));
    KR_TRY(h2d(ctx, pl.d_send_idx, sidx.data(), sizeof(int32_t) * sidx.size()));
    KR_HIP(hipMalloc(&pl.d_sendbuf, sizeof(double) * (size_t)(pl.total_send + 1)));
    return KRYST_OK;
}

// All ranks learn whether any of them failed so far (one all-gather of the status words): 0 if none did, else this rank's own
// code, or KRYST_ERR_ARG with a message naming the first failed rank.
// one int64 from every rank (all[p] = rank p's word); collective
static int32_t all_gather_word(kryst_ctx_t ctx, int64_t mine, std::vector<int64_t>& all) {
    const int P = ctx->nranks;
    all.assign((size_t)P, 0);
    int64_t *d_s = nullptr, *d_r = nullptr;
    int32_t rc = KRYST_OK;
    if (hipMalloc(&d_s, sizeof(int64_t)) != hipSuccess || hipMalloc(&d_r, sizeof(int64_t) * P) != hipSuccess) rc = KRYST_ERR_HIP;
    if (rc == KRYST_OK) rc = h2d(ctx, d_s, &mine, sizeof(int64_t));
    if (rc == KRYST_OK) rc = comm_all_gather_i64(ctx, d_s, d_r, 1, ctx->s_main);
    if (rc == KRYST_OK && (hipMemcpyAsync(all.data(), d_r, sizeof(int64_t) * P, hipMemcpyDeviceToHost, ctx->s_main) != hipSuccess ||
                           hipStreamSynchronize(ctx->s_main) != hipSuccess)) rc = KRYST_ERR_HIP;
    (void)hipFree(d_s); (void)hipFree(d_r);
    return rc;
}

static int32_t agree_on_status(kryst_ctx_t ctx, int32_t rc_mine) {
    const int P = ctx->nranks;
    if (P == 1) return rc_mine;
    std::vector<int64_t> all;
    const int32_t rc = all_gather_word(ctx, rc_mine, all);
    if (rc != KRYST_OK) return rc_mine != KRYST_OK ? rc_mine : rc;
    if (rc_mine != KRYST_OK) return rc_mine;
    for (int p = 0; p < P; ++p)
        if (all[p] != KRYST_OK) { set_error("csr_create_dist failed on rank %d (status %lld)", p, (long long)all[p]); return KRYST_ERR_ARG; }
    return KRYST_OK;
}

int32_t kryst_csr_create_dist(kryst_ctx_t ctx, int64_t n_global, const int64_t* row_offsets, const int64_t* row_ptr,
                              const int64_t* col_global, const double* vals, kryst_csr_t* out) {
    // collective over the context's ranks: a rank whose arguments are bad must still take part in the status agreement
    // below, or the others would wait for it inside the index-list exchange forever
    KR_ARG(ctx && out, "csr_create_dist");
    kryst_csr_t a = nullptr;
    int32_t rc = KRYST_OK;
    if (!(row_offsets && row_ptr)) { set_error("bad argument: csr_create_dist"); rc = KRYST_ERR_ARG; }
    if (rc == KRYST_OK && hipSetDevice(ctx->device) != hipSuccess) { set_error("hipSetDevice failed"); rc = KRYST_ERR_HIP; }
    if (rc == KRYST_OK) {
        a = new kryst_csr_s();
        a->ctx = ctx;
        rc = create_dist_local(a, n_global, row_offsets, row_ptr, col_global, vals);
    }
    rc = agree_on_status(ctx, rc);
    if (rc == KRYST_OK) rc = agree_on_status(ctx, create_dist_exchange(a));
    if (rc == KRYST_OK) {
        // An exchange of the solvers' direction vector that is started early (solvers.hip: launch_direction) adds one exchange per
        // iteration on the rank that starts it: EVERY rank must take that decision alike, so it is taken here, once, from what all ranks
        // report (send lists that are contiguous runs on every rank), never from a rank's own tile counts.
        std::vector<int64_t> all;
        rc = ctx->nranks > 1 ? all_gather_word(ctx, a->send_contiguous ? 1 : 0, all) : KRYST_OK;
        bool every = a->send_contiguous;
        for (int64_t v : all) every = every && v == 1;
        a->halo_early_ok = rc == KRYST_OK && every;
    }
    if (rc == KRYST_OK) rc = halo_default_mode(a);       // (every rank is here with rc == KRYST_OK, or none is)
    if (rc != KRYST_OK) { if (a) kryst_csr_destroy(a); return rc; }
    *out = a;
    return KRYST_OK;
}

// ---- device-side generation of the synthetic 7-point operators (no host arrays, no PCIe) ----
// global prefix of the row lengths: 7*row minus the neighbours that fall outside the grid in rows < row
__host__ __device__ static inline int64_t stencil_gptr(int64_t row, int64_t N) {
    const int64_t N2 = N * N;
    const int64_t plane = row / N2, inpl = row % N2;
    const int64_t f_bottom = row < N2 ? row : N2;                                  // rows with k == 0
    const int64_t f_top = row > (N - 1) * N2 ? row - (N - 1) * N2 : 0;             // rows with k == N-1
    const int64_t f_south = plane * N + (inpl < N ? inpl : N);                     // j == 0
    const int64_t f_north = plane * N + (inpl > (N - 1) * N ? inpl - (N - 1) * N : 0);   // j == N-1
    const int64_t f_west = (row + N - 1) / N;                                      // i == 0
    const int64_t f_east = row / N;                                                // i == N-1
    return 7 * row - (f_bottom + f_top + f_south + f_north + f_west + f_east);
}

struct StencilCoef { double c[7]; int varcoef; };

// kind 3, "varcoef": a symmetric variable-coefficient diffusion operator -- no two rows alike, so no value dictionary and no row
// patterns apply (the operator every structured grid with real coefficients looks like).  The edge between rows r and r + off
// (off = 1, N, N^2: direction d = 0, 1, 2) carries the weight w(r, d) = 0.5 + U(splitmix64(seed 0xD1FF, counter 3 r + d)) in
// [0.5, 1.5); a_{r, r+off} = a_{r+off, r} = -w; the diagonal is the sum, in direction order from 0.0, of the six incident weights
// with 1.0 for a neighbour outside the box (Dirichlet by truncation): a weakly diagonally dominant SPD M-matrix.  The same
// function in ctx.cpp (kryst_host_stencil7) and oracle/oracle.py (stencil7 "varcoef").
__host__ __device__ static inline double varcoef_weight(int64_t r, int d) {
    uint64_t z = 0xD1FFull + ((uint64_t)(3 * r + d) + 1) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z = z ^ (z >> 31);
    return 0.5 + (double)(z >> 11) * (1.0 / 9007199254740992.0);
}

__global__ void stencil7_gen_kernel(int32_t N, int64_t lo, int64_t hi, int64_t n_lower, StencilCoef sc,
                                    int32_t* row_ptr, int32_t* col, double* val, uint8_t* code, uint16_t* code16, uint16_t* pid) {
    const int64_t nloc = hi - lo;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i > nloc) return;
    const int64_t row = lo + i;
    const int64_t base = stencil_gptr(lo, N);
    int64_t k = stencil_gptr(row, N) - base;
    row_ptr[i] = (int32_t)k;
    if (i == nloc) return;
    const int64_t N1 = N, N2 = N1 * N1;
    const int64_t ii = row % N1, jj = (row / N1) % N1, kk = row / N2;
    const bool ok[7] = {kk > 0, jj > 0, ii > 0, true, ii < N1 - 1, jj < N1 - 1, kk < N1 - 1};
    const int64_t off[7] = {-N2, -N1, -1, 0, 1, N1, N2};
    // CSR-P16 pattern id: which neighbours exist (bits 0-5, direction order without the centre) and whether the k
    // neighbours live in a halo plane (bits 6, 7); the table is built on the host (stencil_pattern_table)
    if (pid) pid[i] = (uint16_t)((ok[0] ? 1 : 0) | (ok[1] ? 2 : 0) | (ok[2] ? 4 : 0) | (ok[4] ? 8 : 0) | (ok[5] ? 16 : 0) | (ok[6] ? 32 : 0) |
                        ((ok[0] && row - N2 < lo) ? 64 : 0) | ((ok[6] && row + N2 >= hi) ? 128 : 0));
    double cv[7];
#pragma unroll
    for (int s = 0; s < 7; ++s) cv[s] = sc.c[s];
    if (sc.varcoef) {
        const double w[7] = {ok[0] ? varcoef_weight(row - N2, 2) : 1.0, ok[1] ? varcoef_weight(row - N1, 1) : 1.0, ok[2] ? varcoef_weight(row - 1, 0) : 1.0, 0.0,
                             ok[4] ? varcoef_weight(row, 0) : 1.0, ok[5] ? varcoef_weight(row, 1) : 1.0, ok[6] ? varcoef_weight(row, 2) : 1.0};
        double dsum = 0.0;
#pragma unroll
        for (int s = 0; s < 7; ++s) if (s != 3) { dsum = dsum + w[s]; cv[s] = -w[s]; }
        cv[3] = dsum;
    }
#pragma unroll
    for (int s = 0; s < 7; ++s)
        if (ok[s]) {
            const int64_t c = row + off[s];
            int64_t lc;
            int cd = s;                                             // dictionary: 0..6 in-block offsets
            if (c >= lo && c < hi) lc = c - lo;
            else if (c < lo) { lc = nloc + (c - (lo - N2)); cd = 7; }          // halo plane from rank-1: lc - i == nloc
            else { lc = nloc + n_lower + (c - hi); cd = 8; }                   // halo plane from rank+1: lc - i == n_lower + N^2
            col[k] = (int32_t)lc; val[k] = cv[s]; code[k] = (uint8_t)cd;
            if (code16) code16[k] = (uint16_t)((s << 8) | cd);      // value dictionary = the 7 coefficients, by direction
            ++k;
        }
}

static int32_t create_stencil7_device(kryst_ctx_t ctx, int32_t N, int32_t kind, kryst_csr_t* out) {
    const int P = ctx->nranks, me = ctx->rank;
    const int64_t N2 = (int64_t)N * N, n = N2 * N;
    std::vector<int64_t> offs((size_t)P + 1);
    KR_TRY(kryst_host_partition_rows(n, P, N2, offs.data()));
    const int64_t lo = offs[me], hi = offs[me + 1], nloc = hi - lo;
    const int64_t nnz = stencil_gptr(hi, N) - stencil_gptr(lo, N);
    KR_ARG(nloc < (1ll << 31) - KR_TILE && nnz < (1ll << 31) - 16, "local block exceeds int32 device indexing");
    const bool dist = use_collectives(ctx);
    const bool has_lower = dist && me > 0 && nloc > 0, has_upper = dist && me < P - 1 && nloc > 0;
    KR_ARG(!dist || P == 1 || nloc >= N2, "stencil7: more ranks than grid planes");
    KR_HIP(hipSetDevice(ctx->device));
    kryst_csr_t a = new kryst_csr_s();
    a->ctx = ctx; a->nrows = nloc; a->ncols = n; a->nnz = nnz; a->dist = dist; a->xlen = dist ? nloc : n;
    a->row_offsets = offs;
    StencilCoef sc;
    const bool varcoef = kind == 3;
    sc.varcoef = varcoef ? 1 : 0;
    {   // same coefficients as kryst_host_stencil7
        double* c = sc.c;
        for (int u = 0; u < 7; ++u) c[u] = 0.0;
        if (kind == 0) { c[0] = c[1] = c[2] = c[4] = c[5] = c[6] = -1.0; c[3] = 6.0; }
        else if (kind == 1) { const double cx = 1.0, cy = 1.0, cz = 0.01; c[2] = c[4] = -cx; c[1] = c[5] = -cy; c[0] = c[6] = -cz; c[3] = 2.0 * (cx + cy + cz); }
        else if (kind == 2) { const double gx = 1.0, gy = 0.5, gz = 0.25; c[2] = -(1.0 + gx); c[4] = -1.0; c[1] = -(1.0 + gy); c[5] = -1.0; c[0] = -(1.0 + gz); c[6] = -1.0; c[3] = 6.0 + gx + gy + gz; }
    }
    int32_t rc = KRYST_OK;
    do {
        if (hipMalloc(&a->d_row_ptr, sizeof(int32_t) * (size_t)(nloc + 1 + 8)) != hipSuccess ||
            hipMalloc(&a->d_col, sizeof(int32_t) * (size_t)(nnz + 8)) != hipSuccess ||
            hipMalloc(&a->d_val, sizeof(double) * (size_t)(nnz + 8)) != hipSuccess ||
            hipMalloc(&a->d_code, (size_t)(nnz + 32)) != hipSuccess || hipMalloc(&a->d_dict, sizeof(int32_t) * 256) != hipSuccess ||
            (!varcoef && (hipMalloc(&a->d_code16, sizeof(uint16_t) * (size_t)(nnz + 32)) != hipSuccess || hipMalloc(&a->d_vdict, sizeof(double) * 256) != hipSuccess ||
            hipMalloc(&a->d_pid, sizeof(uint16_t) * (size_t)((nloc + KR_TILE - 1) / KR_TILE * KR_TILE + KR_TILE)) != hipSuccess ||
            hipMalloc(&a->d_pmeta, sizeof(uint32_t) * 2 * KR_PMAX) != hipSuccess || hipMalloc(&a->d_poff, sizeof(int32_t) * (KR_TMAX + 8)) != hipSuccess ||
            hipMalloc(&a->d_pval, sizeof(double) * (KR_TMAX + 8)) != hipSuccess))) { set_error("hipMalloc failed (stencil7)"); rc = KRYST_ERR_HIP; break; }
        int32_t dict[256] = {0};
        const int32_t offs7[7] = {(int32_t)-N2, -N, -1, 0, 1, N, (int32_t)N2};
        for (int u = 0; u < 7; ++u) dict[u] = offs7[u];
        dict[7] = (int32_t)nloc;                                   // lower halo slot - local row
        dict[8] = (int32_t)((has_lower ? N2 : 0) + N2);            // upper halo slot - local row
        (void)hipMemcpyAsync(a->d_dict, dict, sizeof dict, hipMemcpyHostToDevice, ctx->s_main);
        (void)hipStreamSynchronize(ctx->s_main);
        (void)hipMemsetAsync(a->d_code + nnz, 0, 32, ctx->s_main);
        if (!varcoef) {                                            // value dictionary + row patterns: constant coefficients only
            double vd[256] = {0.0};
            for (int u = 0; u < 7; ++u) vd[u] = sc.c[u];
            (void)hipMemcpyAsync(a->d_vdict, vd, sizeof vd, hipMemcpyHostToDevice, ctx->s_main);
            (void)hipStreamSynchronize(ctx->s_main);
            (void)hipMemsetAsync(a->d_code16 + nnz, 0, sizeof(uint16_t) * 32, ctx->s_main);
            // pattern table for the 256 ids of stencil7_gen_kernel: one 7-entry base per (lower halo, upper halo)
            // combination, presence mask from the id's neighbour bits (ids that cannot occur keep length 0)
            std::vector<uint32_t> meta(512, 0u); std::vector<int32_t> poff; std::vector<double> pval;
            uint32_t base_start[4];
            for (int hb = 0; hb < 4; ++hb) {
                base_start[hb] = (uint32_t)poff.size();
                for (int u = 0; u < 7; ++u) {
                    int32_t o = offs7[u];
                    if (u == 0 && (hb & 1)) o = dict[7];
                    if (u == 6 && (hb & 2)) o = dict[8];
                    poff.push_back(o); pval.push_back(sc.c[u]);
                }
            }
            for (int id = 0; id < 256; ++id) {
                if (((id & 64) && !(id & 1)) || ((id & 128) && !(id & 32))) continue;
                const uint32_t mask = ((id & 1) ? 1u : 0u) | ((id & 2) ? 2u : 0u) | ((id & 4) ? 4u : 0u) | 8u |
                                      ((id & 8) ? 16u : 0u) | ((id & 16) ? 32u : 0u) | ((id & 32) ? 64u : 0u);
                meta[2 * (size_t)id] = base_start[(id >> 6) & 3] | (7u << 16);
                meta[2 * (size_t)id + 1] = mask;
            }
            (void)hipMemcpyAsync(a->d_pmeta, meta.data(), sizeof(uint32_t) * 512, hipMemcpyHostToDevice, ctx->s_main);
            (void)hipMemcpyAsync(a->d_poff, poff.data(), sizeof(int32_t) * poff.size(), hipMemcpyHostToDevice, ctx->s_main);
            (void)hipMemcpyAsync(a->d_pval, pval.data(), sizeof(double) * pval.size(), hipMemcpyHostToDevice, ctx->s_main);
            (void)hipStreamSynchronize(ctx->s_main);
            a->npat = 256; a->ntab = (int32_t)poff.size(); a->pat_unroll = 7; a->pat_single = true; a->pat_diag3 = true;
            a->pat_stage_n = (N >= 8 && N <= 1024 && N % 2 == 0) ? N : 0;
            a->pat_far_uniform = a->pat_stage_n > 0 && !dist; a->pat_far_lo = (int32_t)-N2; a->pat_far_hi = (int32_t)N2;
            a->pat_far_interior = a->pat_stage_n > 0;                // rows of INTERIOR tiles (no halo columns) all have the far offsets -N^2 / +N^2
            (void)hipMemsetAsync(a->d_pid, 0, sizeof(uint16_t) * (size_t)((nloc + KR_TILE - 1) / KR_TILE * KR_TILE + KR_TILE), ctx->s_main);
        }
        (void)hipMemsetAsync(a->d_col + nnz, 0, sizeof(int32_t) * 8, ctx->s_main);
        (void)hipMemsetAsync(a->d_val + nnz, 0, sizeof(double) * 8, ctx->s_main);
        const int64_t nthreads = nloc + 1;
        hipLaunchKernelGGL(stencil7_gen_kernel, dim3((unsigned)((nthreads + 255) / 256)), dim3(256), 0, ctx->s_main, N, lo, hi,
                           has_lower ? N2 : 0, sc, a->d_row_ptr, a->d_col, a->d_val, a->d_code, a->d_code16, a->d_pid);
        if (hipGetLastError() != hipSuccess || hipStreamSynchronize(ctx->s_main) != hipSuccess) { set_error("stencil7 generation failed"); rc = KRYST_ERR_HIP; break; }
        a->ntiles = ntiles_of(nloc);
        a->slots = 7;
        if ((rc = build_dia(a)) != KRYST_OK) break;
        if ((rc = build_tile_order(a)) != KRYST_OK) break;
        if (!dist) break;
        // analytic halo plan of a k-slab partition: one grid plane from each neighbour, sent in place
        HaloPlan& pl = a->plan;
        pl.nranks = P; pl.rank = me; pl.row_lo = lo; pl.row_hi = hi;
        pl.recv_counts.assign(P, 0); pl.recv_off.assign(P, 0); pl.send_counts.assign(P, 0); pl.send_off.assign(P, 0);
        if (has_lower) { pl.recv_counts[me - 1] = N2; pl.recv_off[me - 1] = 0; pl.send_counts[me - 1] = N2; pl.send_off[me - 1] = 0; }
        if (has_upper) { pl.recv_counts[me + 1] = N2; pl.recv_off[me + 1] = has_lower ? N2 : 0;
                         pl.send_counts[me + 1] = N2; pl.send_off[me + 1] = nloc - N2; }
        pl.total_recv = (has_lower ? N2 : 0) + (has_upper ? N2 : 0);
        pl.total_send = pl.total_recv;
        pl.recv_cols.clear();                           // global columns of the halo slots: the plane below, then the plane above
        if (has_lower) for (int64_t c = lo - N2; c < lo; ++c) pl.recv_cols.push_back(c);
        if (has_upper) for (int64_t c = hi; c < hi + N2; ++c) pl.recv_cols.push_back(c);
        a->send_contiguous = true;                      // send_off = first local row of each run
        a->halo_early_ok = true;                        // ... on every rank of a k-slab partition (analytic plan: nothing to agree on)
        std::vector<int32_t> ti, tb;
        for (int64_t q = 0; q < a->ntiles; ++q) {
            const int64_t r0 = q * KR_TILE, r1 = std::min<int64_t>(r0 + KR_TILE, nloc);
            const bool bnd = (has_lower && r0 < N2) || (has_upper && r1 > nloc - N2);
            (bnd ? tb : ti).push_back((int32_t)q);
        }
        a->n_interior = (int64_t)ti.size(); a->n_boundary = (int64_t)tb.size();
        a->interior_first = (!ti.empty() && (int64_t)ti.back() - (int64_t)ti.front() + 1 == (int64_t)ti.size()) ? (int64_t)ti.front() : -1;
        if (hipMalloc(&a->d_tiles_interior, sizeof(int32_t) * (ti.size() + 1)) != hipSuccess ||
            hipMalloc(&a->d_tiles_boundary, sizeof(int32_t) * (tb.size() + 1)) != hipSuccess ||
            hipMalloc(&pl.d_halo, sizeof(double) * (size_t)(pl.total_recv + 2)) != hipSuccess) { set_error("hipMalloc failed (halo)"); rc = KRYST_ERR_HIP; break; }
        if (!ti.empty()) (void)hipMemcpyAsync(a->d_tiles_interior, ti.data(), sizeof(int32_t) * ti.size(), hipMemcpyHostToDevice, ctx->s_main);
        if (!tb.empty()) (void)hipMemcpyAsync(a->d_tiles_boundary, tb.data(), sizeof(int32_t) * tb.size(), hipMemcpyHostToDevice, ctx->s_main);
        (void)hipMemsetAsync(pl.d_halo, 0, sizeof(double) * (size_t)(pl.total_recv + 2), ctx->s_main);
        if (hipStreamSynchronize(ctx->s_main) != hipSuccess) { set_error("halo setup failed"); rc = KRYST_ERR_HIP; }
    } while (0);
    // several ranks: the default halo mode is chosen collectively, so a rank that failed above must say so before the others enter it
    if (dist && P > 1 && ctx->comm) {
        rc = agree_on_status(ctx, rc);
        if (rc == KRYST_OK) rc = halo_default_mode(a);
    }
    if (rc == KRYST_OK && !dist) rc = csr_place(a);
    if (rc != KRYST_OK) { kryst_csr_destroy(a); return rc; }
    *out = a;
    return KRYST_OK;
}

static int32_t create_stencil7_host(kryst_ctx_t ctx, int32_t N, int32_t kind, kryst_csr_t* out) {
    const int P = ctx->nranks;
    const int64_t n = (int64_t)N * N * N;
    std::vector<int64_t> offs((size_t)P + 1);
    KR_TRY(kryst_host_partition_rows(n, P, (int64_t)N * N, offs.data()));
    const int64_t N2 = (int64_t)N * N;
    const int32_t k_lo = (int32_t)(offs[ctx->rank] / N2), k_hi = (int32_t)(offs[ctx->rank + 1] / N2);
    const int64_t nnz = kryst_host_stencil7(N, kind, k_lo, k_hi, nullptr, nullptr, nullptr);
    if (nnz < 0) return KRYST_ERR_ARG;
    const int64_t nloc = offs[ctx->rank + 1] - offs[ctx->rank];
    std::vector<int64_t> rp((size_t)nloc + 1), col((size_t)nnz);
    std::vector<double> val((size_t)nnz);
    kryst_host_stencil7(N, kind, k_lo, k_hi, rp.data(), col.data(), val.data());
    if (P == 1 && !use_collectives(ctx)) return create_local(ctx, n, n, rp.data(), col.data(), val.data(), out);
    return kryst_csr_create_dist(ctx, n, offs.data(), rp.data(), col.data(), val.data(), out);
}

int32_t kryst_csr_create_stencil7(kryst_ctx_t ctx, int32_t N, int32_t kind, kryst_csr_t* out) {
    KR_ARG(ctx && out && N >= 1 && kind >= 0 && kind <= 3, "csr_create_stencil7");
    // KRYST_STENCIL_HOST=1: build through the general host path (kryst_host_stencil7 + csr_create[_dist]); the two
    // paths must give identical operators (tests/test_gpu_0_parity.py)
    if (env_int("KRYST_STENCIL_HOST", 0)) return create_stencil7_host(ctx, N, kind, out);
    return create_stencil7_device(ctx, N, kind, out);
}

int32_t kryst_csr_destroy(kryst_csr_t a) {
    if (!a) return KRYST_OK;
    (void)hipSetDevice(a->ctx->device);
    (void)hipStreamSynchronize(a->ctx->s_main);
    (void)hipStreamSynchronize(a->ctx->s_comm);
    (void)hipFree(a->d_row_ptr); (void)hipFree(a->d_col); (void)hipFree(a->d_val); (void)hipFree(a->d_code); (void)hipFree(a->d_dict); (void)hipFree(a->d_code16); (void)hipFree(a->d_vdict);
    (void)hipFree(a->d_pid); (void)hipFree(a->d_pmeta); (void)hipFree(a->d_poff); (void)hipFree(a->d_pval); (void)hipFree(a->d_dia);
    (void)hipFree(a->d_tiles_interior); (void)hipFree(a->d_tiles_boundary); (void)hipFree(a->d_tile_order);
    (void)hipFree(a->plan.d_send_idx); (void)hipFree(a->plan.d_sendbuf); (void)hipFree(a->plan.d_halo);
    halo_peer_destroy(a);
    delete a;
    return KRYST_OK;
}

int32_t kryst_csr_shape(kryst_csr_t a, int64_t* nrows, int64_t* ncols, int64_t* nnz) {
    KR_ARG(a, "csr_shape");
    if (nrows) *nrows = a->nrows;
    if (ncols) *ncols = a->ncols;
    if (nnz) *nnz = a->nnz;
    return KRYST_OK;
}

int32_t kryst_csr_download(kryst_csr_t a, int64_t* row_ptr, int32_t* col, double* vals) {
    KR_ARG(a, "csr_download");
    KR_HIP(hipSetDevice(a->ctx->device));
    if (row_ptr) {
        std::vector<int32_t> rp((size_t)a->nrows + 1);
        KR_HIP(hipMemcpyAsync(rp.data(), a->d_row_ptr, sizeof(int32_t) * rp.size(), hipMemcpyDeviceToHost, a->ctx->s_main));
        KR_HIP(hipStreamSynchronize(a->ctx->s_main));         // (in stream order behind the device generator that may have written the arrays)
        for (size_t i = 0; i < rp.size(); ++i) row_ptr[i] = rp[i];
    }
    if (col && a->nnz) KR_HIP(hipMemcpyAsync(col, a->d_col, sizeof(int32_t) * (size_t)a->nnz, hipMemcpyDeviceToHost, a->ctx->s_main));
    if (vals && a->nnz) KR_HIP(hipMemcpyAsync(vals, a->d_val, sizeof(double) * (size_t)a->nnz, hipMemcpyDeviceToHost, a->ctx->s_main));
    KR_HIP(hipStreamSynchronize(a->ctx->s_main));
    return KRYST_OK;
}


}  // extern "C"
