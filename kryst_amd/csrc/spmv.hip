// CSR SpMV for gfx950 -- replaces CsrMatrix::spmv / spmv_parallel (src/matrix/sparse.rs:56-67,103-114).
//
// HBM-bound (0.134 flop/B; MFMA is useless here), so the kernels differ in how many bytes describe the matrix -- the operator
// is stored in up to four lossless forms built at creation (csr_create.hip) and the most compact one is streamed:
//   spmv_pattern_kernel  CSR-P16  one 16-bit row-pattern id per ROW        (constant-coefficient grid operators)
//   spmv_dict_kernel     CSR-D16  one 16-bit (value, offset) code per entry
//   spmv_rows_kernel     CSR-D8   raw values + 1-byte column-offset codes  (any structured grid)
//   spmv_wave_kernel     plain    8 B value + 4 B column per entry         (everything else)
// Shared contract:
//  * a workgroup owns ROW TILES of KR_TILE = 512 consecutive rows (thread t: rows 2t, 2t+1); every lane sums its rows in
//    ASCENDING column order starting from 0.0 with separate mul and add -- bit-identical to the reference's per-row loop
//    (sparse.rs:107-113) in every form.
//  * the window kernels (wave / rows / dict): each of the 4 waves streams the contiguous nnz range of ITS 128 rows through a
//    private LDS window with coalesced loads, every load of the window in flight before the first use; rows longer than a
//    window continue their running sum in the next one.  Same-wave LDS traffic is ordered by the hardware: NO workgroup barrier.
//  * y is written 16 B per lane; optional fused inner products (d.y, y.y) reuse the tile's rows and produce one partial per
//    tile in the library-wide association order (see ew.h), so CG's (p,Ap) costs no extra pass.
//  * tile -> workgroup map: consecutive tiles (pattern kernel: runs of 8 tiles) round-robin over the 8 XCDs.
//  * distributed: interior tiles run while the halo planes travel over xGMI on a second stream; boundary tiles follow.
//    Column indices >= nloc address the halo buffer.
#include "csr.h"
#include "ew.h"
#include <algorithm>


namespace kr {

typedef int    v2i __attribute__((ext_vector_type(2)));
typedef double v2d __attribute__((ext_vector_type(2)));

struct SpmvArgs {
    const int32_t* row_ptr; const int32_t* col; const double* val;
    const double* x; const double* halo; int32_t nloc;
    double* y; const int32_t* tiles; int32_t ntiles; int32_t nrows;
    const double* dvec; double* partials; int64_t pstride;
    const int* done; int32_t xcd_chunk; int32_t swizzle; int32_t group;
    const uint8_t* code; const int32_t* dict;      // CSR-D8: col = row + dict[code] (nullptr when not compressed)
    const uint16_t* code16; const double* vdict;   // CSR-D16: col = row + dict[c & 255], val = vdict[c >> 8]
    const uint16_t* pid; const uint32_t* pmeta; const int32_t* poff; const double* pval; int32_t npat, ntab;   // CSR-P16
    int32_t pat_red_off;                           // byte offset of the reduction scratch in the kernel's dynamic LDS
    int32_t nloc8;                                 // 8 * nloc
    int32_t tpw;                                   // pattern kernel: consecutive tile slots per workgroup
    int32_t amask;                                 // window start = k0 & ~amask (1: value pairs; 31: whole memory lines)
    const double* dia; int64_t dia_stride; int32_t dia_nd; int32_t dia_min;    // CSR-DIA: value streams, diagonals, lowest offset
    int64_t xsafe; int32_t cmax;                   // last safe start of a 16-byte pair in x's allocation; last valid local column
    int32_t dia_off[KR_DIA_MAX];
#ifdef KR_TUNING
    int32_t abl;                                   // timing-only ablation mask (tuning builds)
#endif
};

template <bool NT, class T>
__device__ __forceinline__ T stream_load(const T* p) {
    if constexpr (NT) return __builtin_nontemporal_load(p);
    else return *p;
}

template <bool HALO>
__device__ __forceinline__ double gather(const SpmvArgs& a, int32_t c) {
    if constexpr (HALO) {
        const double* base = (c < a.nloc) ? a.x : (a.halo - a.nloc);
        return base[c];
    } else {
        return a.x[c];
    }
}

// ascending serial sum of prod[beg-base .. end-base) continued into s: LDS reads are issued four at a time
// (clamped, branch-free) and folded in index order
__device__ __forceinline__ double row_sum(const double* prod, int base, int beg, int end, double s) {
    for (int k = beg; k < end; k += 4) {
        const int last = end - 1 - base;
        const double v0 = prod[k - base];
        const double v1 = prod[min(k + 1 - base, last)];
        const double v2 = prod[min(k + 2 - base, last)];
        const double v3 = prod[min(k + 3 - base, last)];
        s = s + v0;
        if (k + 1 < end) s = s + v1;
        if (k + 2 < end) s = s + v2;
        if (k + 3 < end) s = s + v3;
    }
    return s;
}

// Products-in-LDS form (plain int32 columns): each of the 4 waves streams the nnz range of ITS 128 rows (lane l: rows
// 2l, 2l+1 of the wave's slice) through a private LDS window, gathering x and storing the PRODUCTS.  Same-wave LDS traffic is
// ordered by the hardware, so the main path has NO workgroup barrier: waves run their load / gather / LDS / sum
// phases out of step and hide each other's latency.  Only the optional fused inner product meets at one barrier.
template <int NQ, bool HALO, int SLOTS, bool NT, int MINW = 1>
__global__ __launch_bounds__(KR_T, MINW) void spmv_wave_kernel(const SpmvArgs a) {
    if (a.done && *a.done) return;
    constexpr int WCAP = SLOTS * 128;                       // entries per wave window
    __shared__ __attribute__((aligned(16))) double prod_all[4 * WCAP];
    __shared__ double red[(NQ > 0 ? NQ : 1) * (KR_T / 64)];
    const int t = threadIdx.x, l = t & 63;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);
    double* prod = prod_all + w * WCAP;
    const int xcd = blockIdx.x & 7, slot0 = blockIdx.x >> 3, per = gridDim.x >> 3;
    for (int li = slot0; li < a.xcd_chunk; li += per) {
        int ti;
        if (a.swizzle) ti = xcd * a.xcd_chunk + li;
        else ti = ((li / a.group) * 8 + xcd) * a.group + (li % a.group);
        if (ti >= a.ntiles) { if (a.swizzle) break; else continue; }
        const int q = a.tiles ? a.tiles[ti] : ti;
        if (q < 0) continue;                                    // an empty slot of the slab order
        const int r0 = q * KR_TILE;
        const int r1 = min(r0 + KR_TILE, a.nrows);
        const int wr0 = min(r0 + 128 * w, r1), wr1 = min(wr0 + 128, r1);
        const int row = r0 + 2 * t;
        const int p0 = a.row_ptr[min(row, r1)];
        const int p1 = a.row_ptr[min(row + 1, r1)];
        const int p2 = a.row_ptr[min(row + 2, r1)];
        const int k0 = a.row_ptr[wr0], k1 = a.row_ptr[wr1];         // wave-uniform
        double s0 = 0.0, s1 = 0.0;
#ifdef KR_TUNING
        if (a.abl & 128) { if (row + 1 < r1) st2(a.y, row, (double)p0, (double)p1); }     // the y traffic at the START of the tile (timing only)
#endif
        // a window starts on a memory line of both streams when amask = 31 (32 entries = 128 B of columns, 256 B of values):
        // every load instruction then covers whole lines, which matters for nontemporal loads (no L1 copy of a shared line)
        for (int base = k0 & ~a.amask; base < k1; base += WCAP) {
            const int wend = min(base + WCAP, k1);
            const int npairs = (wend - base + 1) >> 1;
            v2i c[SLOTS]; v2d v[SLOTS]; double xa[SLOTS], xb[SLOTS];
#pragma unroll
            for (int j = 0; j < SLOTS; ++j) {
                const int pi = min(l + j * 64, npairs - 1);
                const int k = base + 2 * pi;
                c[j] = stream_load<NT>(reinterpret_cast<const v2i*>(a.col + k));
                v[j] = stream_load<NT>(reinterpret_cast<const v2d*>(a.val + k));
            }
#ifdef KR_TUNING
            // timing-only ablations (round-2 study, profiles/r02/plain_csr_study/; wrong results): 1 no x gathers, 2 no LDS products / row sums, 4 no y / dot
            if (a.abl & 1) {
#pragma unroll
                for (int j = 0; j < SLOTS; ++j) { xa[j] = (double)c[j].x; xb[j] = (double)c[j].y; }
            } else
#endif
            {
#pragma unroll
            for (int j = 0; j < SLOTS; ++j) { xa[j] = gather<HALO>(a, c[j].x); xb[j] = gather<HALO>(a, c[j].y); }
            }
#ifdef KR_TUNING
            if (a.abl & 2) {
#pragma unroll
                for (int j = 0; j < SLOTS; ++j) { s0 = s0 + v[j].x * xa[j]; s1 = s1 + v[j].y * xb[j]; }
                continue;
            }
#endif
#pragma unroll
            for (int j = 0; j < SLOTS; ++j)
                *reinterpret_cast<double2*>(&prod[2 * (l + j * 64)]) = make_double2(v[j].x * xa[j], v[j].y * xb[j]);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            s0 = row_sum(prod, base, max(p0, base), min(p1, wend), s0);
            s1 = row_sum(prod, base, max(p1, base), min(p2, wend), s1);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
#ifdef KR_TUNING
        if (a.abl & 4) { if (s0 + s1 == 0.123456789) a.y[row] = s0; continue; }
        if (a.abl & 64) { if (row + 1 < r1) st2_keep(a.y, row, s0, s1); else if (row < r1) a.y[row] = s0; } else
#endif
        { if (row + 1 < r1) st2(a.y, row, s0, s1);
        else if (row < r1) a.y[row] = s0; }
        if constexpr (NQ > 0) {
            double acc[NQ];
#ifdef KR_TUNING
            d2 d;
            if (a.abl & 16) { d.a = s1; d.b = s0; }
            else if (a.abl & 8) d = ld2_keep(a.dvec, row);
            else d = ld2(a.dvec, row);
#else
            const d2 d = ld2(a.dvec, row);
#endif
            acc[0] = 0.0;
            if (row < r1) acc[0] = acc[0] + d.a * s0;
            if (row + 1 < r1) acc[0] = acc[0] + d.b * s1;
            if constexpr (NQ > 1) {
                acc[1] = 0.0;
                if (row < r1) acc[1] = acc[1] + s0 * s0;
                if (row + 1 < r1) acc[1] = acc[1] + s1 * s1;
            }
#ifdef KR_TUNING
            if (a.abl & 32) {
#pragma unroll
                for (int k = 0; k < NQ; ++k) acc[k] = wave_butterfly(acc[k]);
                if (l == 0) {
#pragma unroll
                    for (int k = 0; k < NQ; ++k) a.partials[k * a.pstride + q] = acc[k];
                }
                continue;
            }
#endif
            block_reduce<NQ, KR_T / 64>(acc, red);
            if (t == 0) {
#pragma unroll
                for (int k = 0; k < NQ; ++k) a.partials[k * a.pstride + q] = acc[k];
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// "rows" form: phase 1 only STREAMS the matrix (values + column indices, or values + 1-byte column codes) into the
// wave's LDS window -- no dependent gather, every load of the window in flight at once; phase 2 walks each row in
// ascending column order, and the row's owner lane gathers x itself: neighbouring lanes own neighbouring rows, so
// for banded / stencil matrices one gather instruction touches a contiguous run of x (8 cache lines, all used by
// the lane's second row) instead of the 14 scattered lines of an entry-order gather.
//
// CSR-D8 index compression (COMP): when the operator has at most 256 distinct (col - row) offsets -- every
// structured-grid discretisation, whatever its coefficients -- kryst_csr_create stores one byte per entry, a code
// into a 256-entry offset dictionary, beside the plain int32 columns.  The kernel then moves 9 instead of 12 bytes
// per nonzero; the arithmetic (values, ascending column order, un-fused mul/add) is unchanged, so results are
// bit-identical to the plain path.  Matrices with more distinct offsets use the plain int32 columns.
template <int NQ, bool HALO, int SLOTS, bool COMP>
__global__ __launch_bounds__(KR_T) void spmv_rows_kernel(const SpmvArgs a) {
    if (a.done && *a.done) return;
    constexpr int WCAP = SLOTS * 128;                       // entries per wave window
    constexpr int CCAP = WCAP + 32;                         // code bytes per wave window (16-byte aligned start + slack)
    __shared__ __attribute__((aligned(16))) double val_all[4 * WCAP];
    __shared__ __attribute__((aligned(16))) int32_t col_all[COMP ? 4 : 4 * WCAP];
    __shared__ __attribute__((aligned(16))) uint8_t code_all[COMP ? 4 * CCAP : 16];
    __shared__ int32_t dict[COMP ? 256 : 1];
    __shared__ double red[(NQ > 0 ? NQ : 1) * (KR_T / 64)];
    const int t = threadIdx.x, l = t & 63;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);
    double* lval = val_all + w * WCAP;
    int32_t* lcol = col_all + (COMP ? 0 : w * WCAP);
    uint8_t* lcode = code_all + (COMP ? w * CCAP : 0);
    if constexpr (COMP) { dict[t] = a.dict[t]; __syncthreads(); }
    const int xcd = blockIdx.x & 7, slot0 = blockIdx.x >> 3, per = gridDim.x >> 3;
    for (int li = slot0; li < a.xcd_chunk; li += per) {
        int ti;
        if (a.swizzle) ti = xcd * a.xcd_chunk + li;
        else ti = ((li / a.group) * 8 + xcd) * a.group + (li % a.group);
        if (ti >= a.ntiles) { if (a.swizzle) break; else continue; }
        const int q = a.tiles ? a.tiles[ti] : ti;
        if (q < 0) continue;                                    // an empty slot of the slab order
        const int r0 = q * KR_TILE;
        const int r1 = min(r0 + KR_TILE, a.nrows);
        const int wr0 = min(r0 + 128 * w, r1), wr1 = min(wr0 + 128, r1);
        const int row = r0 + 2 * t;
        const int p0 = a.row_ptr[min(row, r1)];
        const int p1 = a.row_ptr[min(row + 1, r1)];
        const int p2 = a.row_ptr[min(row + 2, r1)];
        const int k0 = a.row_ptr[wr0], k1 = a.row_ptr[wr1];         // wave-uniform
        double s0 = 0.0, s1 = 0.0;
        for (int base = k0 & ~1; base < k1; base += WCAP) {
            const int wend = min(base + WCAP, k1);
            const int npairs = (wend - base + 1) >> 1;
            // ---- phase 1: stream the window into LDS
            v2d v[SLOTS];
            [[maybe_unused]] v2i c[SLOTS];
#pragma unroll
            for (int j = 0; j < SLOTS; ++j) {
                const int pi = min(l + j * 64, npairs - 1);
                const int k = base + 2 * pi;
                v[j] = *reinterpret_cast<const v2d*>(a.val + k);
                if constexpr (!COMP) c[j] = *reinterpret_cast<const v2i*>(a.col + k);
            }
            int cbase = 0;
            if constexpr (COMP) {
                cbase = base & ~15;                                   // 16-byte aligned start of the code window
                const int nbytes = wend - cbase;
                const int off = min(16 * l, ((nbytes + 15) & ~15) - 16);
                const uint4 cw = *reinterpret_cast<const uint4*>(a.code + cbase + off);
                if (16 * l < nbytes + 16) *reinterpret_cast<uint4*>(lcode + off) = cw;
            }
#pragma unroll
            for (int j = 0; j < SLOTS; ++j) {
                *reinterpret_cast<double2*>(&lval[2 * (l + j * 64)]) = make_double2(v[j].x, v[j].y);
                if constexpr (!COMP) *reinterpret_cast<int2*>(&lcol[2 * (l + j * 64)]) = make_int2(c[j].x, c[j].y);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            // ---- phase 2: the two rows of the lane, four entries of each in flight, folded in ascending column order
            int ka = max(p0, base), kb = max(p1, base);
            const int ea = min(p1, wend), eb = min(p2, wend);
            while (ka < ea || kb < eb) {
                double va[4], vb[4], xa[4], xb[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int ia = min(ka + u, max(ea - 1, base)), ib = min(kb + u, max(eb - 1, base));
                    int ca, cb;
                    if constexpr (COMP) { ca = row + dict[lcode[ia - cbase]]; cb = row + 1 + dict[lcode[ib - cbase]]; }
                    else { ca = lcol[ia - base]; cb = lcol[ib - base]; }
                    va[u] = lval[ia - base]; vb[u] = lval[ib - base];
                    xa[u] = (ka + u < ea) ? gather<HALO>(a, ca) : 0.0;
                    xb[u] = (kb + u < eb) ? gather<HALO>(a, cb) : 0.0;
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (ka + u < ea) s0 = s0 + va[u] * xa[u];
                    if (kb + u < eb) s1 = s1 + vb[u] * xb[u];
                }
                ka += 4; kb += 4;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        if (row + 1 < r1) st2(a.y, row, s0, s1);
        else if (row < r1) a.y[row] = s0;
        if constexpr (NQ > 0) {
            double acc[NQ];
            const d2 d = ld2(a.dvec, row);
            acc[0] = 0.0;
            if (row < r1) acc[0] = acc[0] + d.a * s0;
            if (row + 1 < r1) acc[0] = acc[0] + d.b * s1;
            if constexpr (NQ > 1) {
                acc[1] = 0.0;
                if (row < r1) acc[1] = acc[1] + s0 * s0;
                if (row + 1 < r1) acc[1] = acc[1] + s1 * s1;
            }
            block_reduce<NQ, KR_T / 64>(acc, red);
            if (t == 0) {
#pragma unroll
                for (int k = 0; k < NQ; ++k) a.partials[k * a.pstride + q] = acc[k];
            }
        }
    }
}

// CSR-D16, the fully dictionary-coded form (spmv_dict_kernel): when, besides the <= 256 distinct (col - row) offsets of
// CSR-D8, the operator also has at most 256 distinct VALUES (bit patterns) -- constant-coefficient discretisations on
// uniform grids: every BASELINE config -- kryst_csr_create keeps one 16-bit word per entry, (value code << 8) | offset
// code, and two 256-entry dictionaries.  The kernel then streams 2 instead of 12 bytes per nonzero (34 instead of 95
// bytes per row of a 7-point stencil, vectors included).  The decoded value is the stored double bit for bit and the
// arithmetic (ascending column order from 0.0, separate mul and add) is unchanged, so y is bit-identical to the plain
// path; operators with more distinct values or offsets use CSR-D8 or plain CSR.  A wave's window is its slice's run
// of code words (<= 1016 per pass), loaded 16 B per lane from an 8-entry-aligned start; no workgroup barrier after
// the dictionaries are in LDS.
template <int NQ, bool HALO>
__global__ __launch_bounds__(KR_T) __attribute__((amdgpu_waves_per_eu(8, 8))) void spmv_dict_kernel(const SpmvArgs a) {
    if (a.done && *a.done) return;
    constexpr int WCAP = 1016;                              // entries per wave pass (2 x 64 lanes x 8 entries, minus alignment slack)
    constexpr int CCAP = 1024 + 8;
    __shared__ __attribute__((aligned(16))) uint16_t code_all[4 * CCAP];
    __shared__ int32_t dict[256];
    __shared__ double vdict[256];
    __shared__ double red[(NQ > 0 ? NQ : 1) * (KR_T / 64)];
    const int t = threadIdx.x, l = t & 63;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);
    uint16_t* lcode = code_all + w * CCAP;
    dict[t] = a.dict[t]; vdict[t] = a.vdict[t];
    __syncthreads();
    const int xcd = blockIdx.x & 7, slot0 = blockIdx.x >> 3, per = gridDim.x >> 3;
    for (int li = slot0; li < a.xcd_chunk; li += per) {
        int ti;
        if (a.swizzle) ti = xcd * a.xcd_chunk + li;
        else ti = ((li / a.group) * 8 + xcd) * a.group + (li % a.group);
        if (ti >= a.ntiles) { if (a.swizzle) break; else continue; }
        const int q = a.tiles ? a.tiles[ti] : ti;
        if (q < 0) continue;                                    // an empty slot of the slab order
        const int r0 = q * KR_TILE;
        const int r1 = min(r0 + KR_TILE, a.nrows);
        const int wr0 = min(r0 + 128 * w, r1), wr1 = min(wr0 + 128, r1);
        const int row = r0 + 2 * t;
        const int p0 = a.row_ptr[min(row, r1)];
        const int p1 = a.row_ptr[min(row + 1, r1)];
        const int p2 = a.row_ptr[min(row + 2, r1)];
        const int k0 = a.row_ptr[wr0], k1 = a.row_ptr[wr1];         // wave-uniform
        double s0 = 0.0, s1 = 0.0;
        for (int base = k0; base < k1; base += WCAP) {
            const int wend = min(base + WCAP, k1);
            // ---- phase 1: the window's code words -> LDS (two 16-byte loads per lane, both in flight)
            const int cbase = base & ~7;
            const int nent = wend - cbase;                            // <= WCAP + 7
            const int lastoff = ((nent + 7) & ~7) - 8;
            const int o0 = min(8 * l, lastoff), o1 = min(8 * (l + 64), lastoff);
            const uint4 c0 = *reinterpret_cast<const uint4*>(a.code16 + cbase + o0);
            const uint4 c1 = *reinterpret_cast<const uint4*>(a.code16 + cbase + o1);
            if (8 * l < nent) *reinterpret_cast<uint4*>(lcode + o0) = c0;
            if (8 * (l + 64) < nent) *reinterpret_cast<uint4*>(lcode + o1) = c1;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            // ---- phase 2: the two rows of the lane, four entries of each in flight, folded in ascending column order
            int ka = max(p0, base), kb = max(p1, base);
            const int ea = min(p1, wend), eb = min(p2, wend);
            while (ka < ea || kb < eb) {
                // all gathers of the batch are issued before the first value is decoded (U = 7: a 7-point row in one trip)
                constexpr int U = 7;
                double xa[U], xb[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int ia = min(ka + u, max(ea - 1, base)), ib = min(kb + u, max(eb - 1, base));
                    const int ca = row + dict[lcode[ia - cbase] & 255u], cb = row + 1 + dict[lcode[ib - cbase] & 255u];
                    xa[u] = (ka + u < ea) ? gather<HALO>(a, ca) : 0.0;
                    xb[u] = (kb + u < eb) ? gather<HALO>(a, cb) : 0.0;
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {                       // the code words are re-read from LDS: cheaper than holding them
                    const int ia = min(ka + u, max(ea - 1, base)), ib = min(kb + u, max(eb - 1, base));
                    if (ka + u < ea) s0 = s0 + vdict[lcode[ia - cbase] >> 8] * xa[u];
                    if (kb + u < eb) s1 = s1 + vdict[lcode[ib - cbase] >> 8] * xb[u];
                }
                ka += U; kb += U;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        if (row + 1 < r1) st2(a.y, row, s0, s1);
        else if (row < r1) a.y[row] = s0;
        if constexpr (NQ > 0) {
            double acc[NQ];
            const d2 d = ld2(a.dvec, row);
            acc[0] = 0.0;
            if (row < r1) acc[0] = acc[0] + d.a * s0;
            if (row + 1 < r1) acc[0] = acc[0] + d.b * s1;
            if constexpr (NQ > 1) {
                acc[1] = 0.0;
                if (row < r1) acc[1] = acc[1] + s0 * s0;
                if (row + 1 < r1) acc[1] = acc[1] + s1 * s1;
            }
            block_reduce<NQ, KR_T / 64>(acc, red);
            if (t == 0) {
#pragma unroll
                for (int k = 0; k < NQ; ++k) a.partials[k * a.pstride + q] = acc[k];
            }
        }
    }
}

// CSR-P16, row-pattern coding (spmv_pattern_kernel): operators assembled from a few stencils -- constant-coefficient
// discretisations on structured grids, i.e. every BASELINE config -- repeat the same ROW, as a sequence of
// (col - row, value) pairs, millions of times (a 7-point operator on a box has 27 distinct rows, all of them
// sub-sequences of the interior row).  kryst_csr_create numbers the distinct rows ("patterns", at most KR_PMAX) and
// expresses each as a BASE sequence (at most KR_TMAX table entries in total) plus a 16-bit presence mask; it keeps one
// 16-bit pattern id per row beside the CSR arrays, and the SpMV streams 2 bytes per ROW (18 bytes per row with x and y,
// against 95 for plain CSR): no row pointers, no per-entry codes, no dependent row_ptr -> entries load chain.  Each lane
// walks its rows' entries in the stored (ascending column) order with the reference's un-fused mul/add, skipping the
// masked-out ones, so y is bit-identical to the plain path.  Everything else falls back to CSR-D16 / CSR-D8 / plain CSR.
//
// At 18 bytes per row the kernel is bound by instruction issue, not by HBM (rocprofv3 --pmc: TA busy 59 %, the texture path
// spends ~16 cycles per 64-lane load whatever its width), so it is built around 16-BYTE gathers: thread t owns the adjacent
// rows 2t, 2t+1; when both rows share a base (interior rows and the boundary rows next to them do) entry e of both rows
// reads x[2t + off_e] and x[2t + 1 + off_e] -- one dwordx4 load -- and one table lookup serves both rows.  Row pairs with
// different bases take the two-loads-per-entry path.  x is addressed with 32-bit byte offsets from a scalar base (the
// launch checks xlen < 2^28).
// x[c8 / 8] for a byte offset c8 >= 0 (local entries below nloc8, halo entries above)
template <bool HALO>
__device__ __forceinline__ const char* gather_base8(const SpmvArgs& a, int32_t c8) {
    if constexpr (HALO) return (c8 < a.nloc8) ? reinterpret_cast<const char*>(a.x) : reinterpret_cast<const char*>(a.halo) - a.nloc8;
    else return reinterpret_cast<const char*>(a.x);
}
template <bool HALO>
__device__ __forceinline__ double gather8(const SpmvArgs& a, int32_t c8) {
    return *reinterpret_cast<const double*>(gather_base8<HALO>(a, c8) + (uint32_t)c8);
}

// SINGLE: no base is longer than U entries (true for every stencil operator), so the entry loop has one trip
// CENTER >= 0: the fused dot's vector is x itself and every row has its diagonal entry at table position CENTER.
// MERGE3: table positions 2, 3, 4 of every base are the columns row-1, row, row+1 (stencil generator): x[row-1 .. row+2] is
// fetched with two 16-byte gathers instead of three (the kernel's cost is per vector-memory instruction).
template <int NQ, bool HALO, int U, bool SINGLE, int CENTER = -1, bool MERGE3 = false>
__global__ __launch_bounds__(KR_T) void spmv_pattern_kernel(const SpmvArgs a) {
    if (a.done && *a.done) return;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double* pval = reinterpret_cast<double*>(smem);                                   // ntab + U entries
    int32_t* poff = reinterpret_cast<int32_t*>(pval + a.ntab + U);                    // BYTE offsets: 8 * (col - row)
    uint2* meta = reinterpret_cast<uint2*>(poff + a.ntab + U + ((a.ntab + U) & 1));   // 8-byte aligned
    double* red = reinterpret_cast<double*>(smem + a.pat_red_off);
    const int t = threadIdx.x;
    // workgroup b of XCD x owns the tpw CONSECUTIVE tile slots [b*tpw, (b+1)*tpw) of that XCD's share; workgroups are
    // handed out in order by the dispatcher, so the tiles in flight on an XCD stay a compact window of the grid and the
    // x planes they share stay in its L2 (a strided persistent loop lets fast workgroups run ahead: 2.5x the x traffic)
    const int xcd = blockIdx.x & 7, slot0 = blockIdx.x >> 3, tpw = a.tpw;
    const int li_end = min((slot0 + 1) * tpw, a.xcd_chunk);
    auto tile_of = [&](int li) -> int {
        if (li >= li_end) return -1;
        const int ti = a.swizzle ? xcd * a.xcd_chunk + li : ((li / a.group) * 8 + xcd) * a.group + (li % a.group);
        if (ti >= a.ntiles) return -1;
        return a.tiles ? a.tiles[ti] : ti;
    };
    int q = tile_of(slot0 * tpw);
    unsigned ids = 0;
    if (q >= 0) ids = *reinterpret_cast<const unsigned*>(a.pid + (size_t)q * KR_TILE + 2 * t);   // rows 2t, 2t+1 (padded array)
    // (the first tile's ids are on their way while the tables are copied)
    for (int i = t; i < a.npat; i += KR_T) meta[i] = reinterpret_cast<const uint2*>(a.pmeta)[i];
    for (int i = t; i < a.ntab + U; i += KR_T) { poff[i] = i < a.ntab ? 8 * a.poff[i] : 0; pval[i] = i < a.ntab ? a.pval[i] : 0.0; }
    __syncthreads();
    for (int li = slot0 * tpw; li < li_end; ++li) {
        const int qn = tile_of(li + 1);                             // prefetch the next tile's ids
        unsigned ids_n = 0;
        if (qn >= 0) ids_n = *reinterpret_cast<const unsigned*>(a.pid + (size_t)qn * KR_TILE + 2 * t);
        if (q >= 0) {
            const int r0 = q * KR_TILE;
            const int r1 = min(r0 + KR_TILE, a.nrows);
            const int row = r0 + 2 * t;
            const int32_t row8 = row << 3;                          // xlen < 2^28: byte offsets fit 31 bits
            const bool va = row < r1, vb = row + 1 < r1;
            uint2 ma = va ? meta[ids & 0xffffu] : make_uint2(0u, 0u), mb = vb ? meta[ids >> 16] : make_uint2(0u, 0u);
            double s0 = 0.0, s1 = 0.0;
            v2d xc; xc.x = 0.0; xc.y = 0.0;                         // x[row], x[row + 1] as the diagonal entry's gather saw them
            if (ma.x == mb.x) {
                // ---- same base: one table lookup and one 16-byte gather per entry serve both rows.  An entry is needed if
                // either row has it; a needed entry addresses x[c], x[c + 1] with c >= -1 (c = -1: only row 2t+1 has it and
                // its column is 0) and c + 1 <= xlen (c + 1 = xlen: only row 2t has it; x is padded at the end).
                const int len = ma.x >> 16;
                const double* tv = pval + (ma.x & 0xffffu); const int32_t* to = poff + (ma.x & 0xffffu);
                for (int e0 = 0; e0 < (SINGLE ? 1 : len); e0 += U) {
                    // masks of the entries e0 .. e0+U-1 (a mask never has bits at or beyond the base length; bases longer than 16
                    // entries have no sub-patterns: all ones up to the length)
                    unsigned ka, kb;
                    if constexpr (SINGLE) { ka = ma.y; kb = mb.y; }
                    else {
                        const unsigned lim = (len - e0 >= 32) ? 0xffffffffu : ((1u << (len - e0)) - 1u);
                        ka = (e0 < 16 ? (ma.y >> e0) : 0xffffffffu) & lim; kb = (e0 < 16 ? (mb.y >> e0) : 0xffffffffu) & lim;
                    }
                    const unsigned kk = ka | kb;
                    int32_t c8[U];
                    int32_t any = 0;
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        c8[u] = ((kk >> u) & 1u) ? row8 + to[e0 + u] : row8;
                        if (MERGE3 && u == 2) c8[u] = row8 - 8;          // always: its second element is x[row]
                        if (MERGE3 && u == 4) c8[u] = row8 + 8;          // always: its first element is x[row + 1] (x is padded at the end)
                        any |= c8[u];
                    }
                    v2d xx[U];
                    if (HALO || __builtin_amdgcn_ballot_w64(any < 0) != 0) {
                        // rare: some lane's pair starts one element before x (or the tile touches halo planes)
#pragma unroll
                        for (int u = 0; u < U; ++u) {
                            xx[u].x = 0.0; xx[u].y = 0.0;
                            if ((ka >> u) & 1u) xx[u].x = gather8<HALO>(a, c8[u]);
                            if ((kb >> u) & 1u) xx[u].y = gather8<HALO>(a, c8[u] + 8);
                        }
                    } else {
#pragma unroll
                        for (int u = 0; u < U; ++u)
                            if (!(MERGE3 && u == 3)) xx[u] = *reinterpret_cast<const v2d*>(reinterpret_cast<const char*>(a.x) + (uint32_t)c8[u]);
                        if constexpr (MERGE3) { xx[3].x = xx[2].y; xx[3].y = xx[4].x; }
                    }
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const double v = tv[e0 + u];
                        const double ta = s0 + v * xx[u].x, tb = s1 + v * xx[u].y;
                        s0 = ((ka >> u) & 1u) ? ta : s0;
                        s1 = ((kb >> u) & 1u) ? tb : s1;
                    }
                    if constexpr (CENTER >= 0) xc = xx[CENTER];
                }
            } else {
                // ---- different bases (rare: adjacent rows built from unrelated stencils): each row walks its own, one entry
                // at a time -- kept deliberately small so that it does not cost the common path registers
                const int la = ma.x >> 16, lb = mb.x >> 16;
                const double* tva = pval + (ma.x & 0xffffu); const int32_t* toa = poff + (ma.x & 0xffffu);
                const double* tvb = pval + (mb.x & 0xffffu); const int32_t* tob = poff + (mb.x & 0xffffu);
#pragma unroll 1
                for (int e = 0; e < max(la, lb); ++e) {
                    const bool ia = e < la && (e >= 16 || ((ma.y >> e) & 1u)), ib = e < lb && (e >= 16 || ((mb.y >> e) & 1u));
                    if (ia) s0 = s0 + tva[e] * gather8<HALO>(a, row8 + toa[e]);
                    if (ib) s1 = s1 + tvb[e] * gather8<HALO>(a, row8 + 8 + tob[e]);
                }
            }
            if (vb) st2(a.y, row, s0, s1);
            else if (va) a.y[row] = s0;
            if constexpr (NQ > 0) {
                double acc[NQ];
                d2 d;
                if (CENTER >= 0 && ma.x == mb.x) { d.a = xc.x; d.b = xc.y; }   // (p, Ap): no second load of p, no second wait
                else d = ld2(a.dvec, row);
                acc[0] = 0.0;
                if (va) acc[0] = acc[0] + d.a * s0;
                if (vb) acc[0] = acc[0] + d.b * s1;
                if constexpr (NQ > 1) {
                    acc[1] = 0.0;
                    if (va) acc[1] = acc[1] + s0 * s0;
                    if (vb) acc[1] = acc[1] + s1 * s1;
                }
                block_reduce<NQ, KR_T / 64>(acc, red);
                if (t == 0) {
#pragma unroll
                    for (int k = 0; k < NQ; ++k) a.partials[k * a.pstride + q] = acc[k];
                }
            }
        }
        q = qn; ids = ids_n;
    }
}

// CSR-P16 with the NEAR operands staged in LDS (spmv_pattern_stage_kernel).  spmv_pattern_kernel pays per vector-memory
// instruction (six 16-byte gathers per tile and wave) and per tile a chain of dependent latencies (ids -> tables -> gathers -> store ->
// fold).  For operators whose every base is (far, -n, -1, 0, +1, +n, far) with one even n <= 1024 (csr.h: pat_stage_n -- a box of lines
// of n points; the generator-made operators and host-built ones alike) a workgroup takes a RUN of T consecutive tiles and
//   * requests, together, the ids of all T tiles and the run's window of x -- x[r0 - n - 2 .. r0 + 512 T + n + 2), T + (2 n + 4) / 512
//     coalesced 16-byte loads per lane instead of 4 T gathers -- and copies the tables; one barrier;
//   * requests the far operands (table positions 0 and 6) of all T tiles together;
//   * then computes tile after tile out of LDS: five aligned 16-byte LDS reads give x[row - n], x[row - 2 .. row + 3], x[row + n].
// Two global round trips per RUN instead of a dependent chain per tile, 2 T + T + (2 n + 4) / 512 global loads instead of 6 T.
// The arithmetic is spmv_pattern_kernel's: every row's entries in stored order, un-fused multiply and add, absent entries skipped by
// a select -- the same bits.
// UFAR: the far offsets (table positions 0 and 6) are the same in every base: the far operands are requested with everything else
// (ONE round trip per run), with clamped addresses where a row has no such entry.
template <int NQ, int T, bool CENTER, bool UFAR>
__global__ __launch_bounds__(KR_T) void spmv_pattern_stage_kernel(const SpmvArgs a, const int32_t n, const int32_t far_lo, const int32_t far_hi, const int32_t tile0) {
    if (a.done && *a.done) return;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int XS = T * KR_TILE + 2 * n + 4;                                  // staged elements (even)
    double* xs = reinterpret_cast<double*>(smem);
    uint2* meta = reinterpret_cast<uint2*>(xs + XS);
    double* pval = reinterpret_cast<double*>(meta + a.npat);
    int32_t* poff = reinterpret_cast<int32_t*>(pval + a.ntab);
    double* red = reinterpret_cast<double*>(smem + a.pat_red_off);
    const int t = threadIdx.x;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int G = a.group;                                                   // G consecutive runs per XCD (neighbouring runs share window edges)
    // runs of T consecutive tiles of the range [tile0, tile0 + ntiles) (the whole operator, or a rank's interior tiles), groups of G
    // runs round-robin over the XCDs
    const int qr = (((slot / G) * 8 + xcd) * G + slot % G) * T;
    if (qr >= a.ntiles) return;                                              // (uniform over the workgroup)
    const int nt = min(T, a.ntiles - qr);
    const int q0 = tile0 + qr;
    const int32_t r0 = q0 * KR_TILE;                                         // (the launch checks xlen + 2 tiles < 2^31: 32-bit element indices)
    // ---- round trip 1: ids of the run, the window of x, the tables
    // The kernel is bound by VALU issue (a 64-lane instruction takes four cycles whatever it does; 770 of them per wave and run
    // before this paragraph, 45 % selects and compares): indices are 32-bit, and a run that lies INSIDE x with its whole window and
    // its far operands -- all but the first and last few of a box -- takes a path without a single clamp (uniform over the workgroup).
    unsigned ids[T];
#pragma unroll
    for (int k = 0; k < T; ++k) ids[k] = k < nt ? *reinterpret_cast<const unsigned*>(a.pid + (size_t)(q0 + k) * KR_TILE + 2 * t) : 0u;
    constexpr int NP = (T * KR_TILE + 2 * 1024 + 4 + 2 * KR_T - 1) / (2 * KR_T);   // pairs per lane at most (n <= 1024)
    const int npairs = XS / 2;
    const int32_t e0 = r0 - n - 2;                                           // element of x staged at xs[0] (even)
    const int32_t xsafe = (int32_t)a.xsafe;
    const bool inside = e0 >= 0 && e0 + XS <= xsafe && (!UFAR || ((int64_t)r0 + far_lo >= 0 && (int64_t)r0 + T * KR_TILE + far_hi <= (int64_t)xsafe));
    // the window goes STRAIGHT into LDS (global_load_lds_dwordx4, gfx950's LDS-DMA: destination = a wave-uniform base + 16 bytes per
    // lane, which is exactly the window's layout): no staging registers, no LDS store instructions
    const int wbase = __builtin_amdgcn_readfirstlane(t & ~63);
    uint2 ma[T], mb[T]; v2d lo[T], hi[T]; d2 dv[T];
    if (inside) {
        const double* xw = a.x + e0 + 2 * t;
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            if (t + i * KR_T < npairs)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(xw + 2 * i * KR_T),
                                                 (__attribute__((address_space(3))) void*)(xs + 2 * (i * KR_T + wbase)), 16, 0, 0);
        }
        if constexpr (UFAR) {
            const double* xl = a.x + (r0 + 2 * t) + far_lo; const double* xh = a.x + (r0 + 2 * t) + far_hi;
#pragma unroll
            for (int k = 0; k < T; ++k) {
#if defined(KR_STAGE_ABL) && (KR_STAGE_ABL & 2)
                lo[k].x = 1.0; lo[k].y = 2.0; hi[k].x = 3.0; hi[k].y = 4.0;  // timing only: no far operands
#else
                lo[k] = *reinterpret_cast<const v2d*>(xl + k * KR_TILE);
                hi[k] = *reinterpret_cast<const v2d*>(xh + k * KR_TILE);
#endif
            }
        }
    } else {
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int pi = t + i * KR_T;
            if (pi < npairs) {
                const int32_t e = min(max(e0 + 2 * pi, 0), xsafe);             // outside x: any valid pair (those operands are absent entries)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a.x + e),
                                                 (__attribute__((address_space(3))) void*)(xs + 2 * (i * KR_T + wbase)), 16, 0, 0);
            }
        }
        if constexpr (UFAR) {
#pragma unroll
            for (int k = 0; k < T; ++k) {
                const int64_t row = (int64_t)r0 + k * KR_TILE + 2 * t;
                lo[k] = *reinterpret_cast<const v2d*>(a.x + min(max(row + far_lo, (int64_t)0), (int64_t)xsafe));
                hi[k] = *reinterpret_cast<const v2d*>(a.x + min(max(row + far_hi, (int64_t)0), (int64_t)xsafe));
            }
        }
    }
    if constexpr (UFAR && NQ > 0 && !CENTER) {
#pragma unroll
        for (int k = 0; k < T; ++k) { dv[k].a = 0.0; dv[k].b = 0.0; if (k < nt) dv[k] = ld2(a.dvec, r0 + k * KR_TILE + 2 * t); }
    }
    for (int i = t; i < a.npat; i += KR_T) meta[i] = reinterpret_cast<const uint2*>(a.pmeta)[i];
    for (int i = t; i < a.ntab; i += KR_T) { poff[i] = a.poff[i]; pval[i] = a.pval[i]; }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                       // the window has landed (the compiler does not count LDS-DMA requests)
    __syncthreads();
    // ---- round trip 2 (!UFAR): the far operands of every tile of the run (and the fused dot's vector when it is not x itself)
#pragma unroll
    for (int k = 0; k < T; ++k) {
        const int32_t row = r0 + k * KR_TILE + 2 * t;
        const bool va = k < nt && row < a.nrows, vb = k < nt && row + 1 < a.nrows;
        ma[k] = va ? meta[ids[k] & 0xffffu] : make_uint2(0u, 0u);
        mb[k] = vb ? meta[ids[k] >> 16] : make_uint2(0u, 0u);
        if constexpr (UFAR) continue;
        const int ba = ma[k].x & 0xffffu, bb = mb[k].x & 0xffffu;
        lo[k].x = 0.0; lo[k].y = 0.0; hi[k].x = 0.0; hi[k].y = 0.0;
        const bool a0 = ma[k].y & 1u, b0 = mb[k].y & 1u, a6 = (ma[k].y >> 6) & 1u, b6 = (mb[k].y >> 6) & 1u;
#if defined(KR_STAGE_ABL) && (KR_STAGE_ABL & 2)
        if (t == 12345) lo[k] = *reinterpret_cast<const v2d*>(a.x + row + poff[ba]);        // timing only: no far operands
        if (t == 12345) hi[k] = *reinterpret_cast<const v2d*>(a.x + row + poff[ba + 6]);
        continue;
#endif
        if (a0 && b0 && ba == bb) lo[k] = *reinterpret_cast<const v2d*>(a.x + row + poff[ba]);
        else { if (a0) lo[k].x = a.x[row + poff[ba]]; if (b0) lo[k].y = a.x[row + 1 + poff[bb]]; }
        if (a6 && b6 && ba == bb) hi[k] = *reinterpret_cast<const v2d*>(a.x + row + poff[ba + 6]);
        else { if (a6) hi[k].x = a.x[row + poff[ba + 6]]; if (b6) hi[k].y = a.x[row + 1 + poff[bb + 6]]; }
        if constexpr (NQ > 0 && !CENTER) { dv[k].a = 0.0; dv[k].b = 0.0; if (k < nt) dv[k] = ld2(a.dvec, row); }
    }
    // ---- tile after tile out of LDS
#pragma unroll
    for (int k = 0; k < T; ++k) {
        if (k >= nt) break;                                                  // (uniform)
        const int32_t row = r0 + k * KR_TILE + 2 * t;
        const bool va = row < a.nrows, vb = row + 1 < a.nrows;
        const int j = k * KR_TILE + 2 * t + n + 2;                           // xs[j] = x[row]
        const v2d A = *reinterpret_cast<const v2d*>(xs + j - 2), B = *reinterpret_cast<const v2d*>(xs + j), C = *reinterpret_cast<const v2d*>(xs + j + 2);
        const v2d M = *reinterpret_cast<const v2d*>(xs + j - n), P = *reinterpret_cast<const v2d*>(xs + j + n);
        v2d e[7];
        e[0] = lo[k]; e[1] = M; e[2].x = A.y; e[2].y = B.x; e[3] = B; e[4].x = B.y; e[4].y = C.x; e[5] = P; e[6] = hi[k];
        const double* tva = pval + (ma[k].x & 0xffffu); const double* tvb = pval + (mb[k].x & 0xffffu);
        const unsigned ka = ma[k].y, kb = mb[k].y;
        double s0 = 0.0, s1 = 0.0;
        // (a scalar branch per entry for the entries every row of the wave has -- no select then -- was tried: 114 branches per run cost
        // more than the selects they save, 0.59 -> 0.64 ms at 512^3; so was hoisting the masks' compares and the table values out of
        // the tile loop when a lane's ids repeat over the run, with the values as scalars when the wave has one base: 486 instead of 770
        // VALU instructions per wave and run, and no faster -- 0.576-0.580 ms, the extra paths cost the kernel a wave per SIMD.
        // rocprofv3 --pmc: VALU busy 47 %, LDS 15 %, 21 % of the wave cycles waiting for memory)
#pragma unroll
        for (int u = 0; u < 7; ++u) {
            const double ta = s0 + tva[u] * e[u].x, tb = s1 + tvb[u] * e[u].y;
            s0 = ((ka >> u) & 1u) ? ta : s0;
            s1 = ((kb >> u) & 1u) ? tb : s1;
        }
#if defined(KR_STAGE_ABL) && (KR_STAGE_ABL & 1)
        if (s0 == 1.23456789e-300) a.y[row] = s0;                           // timing only: no y traffic
#else
        if (vb) st2(a.y, row, s0, s1);
        else if (va) a.y[row] = s0;
#endif
        if constexpr (NQ > 0) {
            // the tile's partials in the library's order (thread: its two rows; wave: butterfly; workgroup: the four waves in
            // turn) -- the cross-wave step of all T tiles of the run meets at ONE barrier below
            double acc[NQ];
            d2 d;
            if constexpr (CENTER) { d.a = B.x; d.b = B.y; } else d = dv[k];
            acc[0] = 0.0;
            if (va) acc[0] = acc[0] + d.a * s0;
            if (vb) acc[0] = acc[0] + d.b * s1;
            if constexpr (NQ > 1) {
                acc[1] = 0.0;
                if (va) acc[1] = acc[1] + s0 * s0;
                if (vb) acc[1] = acc[1] + s1 * s1;
            }
#pragma unroll
            for (int kk = 0; kk < NQ; ++kk) {
                const double wsum = wave_butterfly(acc[kk]);
                if ((t & 63) == 0) red[(k * NQ + kk) * (KR_T / 64) + (t >> 6)] = wsum;
            }
        }
    }
    if constexpr (NQ > 0) {
        __syncthreads();
        if (t < nt * NQ) {                                                   // thread k * NQ + kk: tile q0 + k, quantity kk
            const int k = t / NQ, kk = t % NQ;
            double sum = red[t * (KR_T / 64)];
#pragma unroll
            for (int w2 = 1; w2 < KR_T / 64; ++w2) sum = sum + red[t * (KR_T / 64) + w2];
            a.partials[kk * a.pstride + q0 + k] = sum;
        }
    }
}

// CG / PCG with the direction pass INSIDE the SpMV (spmv_pattern_fuse_kernel; round 5).  An iteration used to end with the
// direction pass  x += alpha p, p = z + beta p  (5 vector words per row) and the next one began with the SpMV reading that p again.  Here the
// staged-window kernel forms p itself: its window fill reads z and p_old instead of p and stores  z + beta p_old  -- cg.rs:274-276 /
// pcg.rs:215-217, the same un-fused multiply and add per element, so the same bits -- into LDS; the rows the run OWNS also go to p_new in
// memory, and the deferred  x += alpha p_old  (cg.rs:207-209) rides on the p_old values the fill holds anyway.  The far operands (rows +-
// far_lo / far_hi away) are formed the same way from z and p_old there: a row's p_new is computed by up to three workgroups, from the same
// operands in the same order.  p_old and p_new are DIFFERENT arrays (the solver ping-pongs): nobody reads what another workgroup writes.
// Per row: z, p_old, x read, p_new, x, y written + the pattern id -- 50 bytes instead of 40 (direction) + 18 (SpMV), and one launch fewer.
//   xpend == it - 1: the x update of the previous iteration is owed (always, unless the solve ended earlier and it has been paid);
//   done: the solve has ended -- only the owed x update happens, nothing else is touched.
struct FuseArgs { const double* z; const double* p_old; double* p_new; double* xvec; const double* alpha; const double* beta; const long long* xpend; long long it; };
// NMAX: the line length the instance is built for (the window's pairs per lane are a compile-time count): 512 or 1024
// XU: the deferred x update rides on this kernel (false: the solver updates x in batches, solvers.hip: XBatchOp -- no registers held for x)
// (measured and not kept, round 5: runs of 8 tiles -- 166 registers, 3 waves per SIMD: 582-590 against 596 it/s; amdgpu_waves_per_eu(5): four
// spilled registers, 565 against 596)
template <int NQ, int T, int NMAX, bool XU>
__global__ __launch_bounds__(KR_T) void spmv_pattern_fuse_kernel(const SpmvArgs a, const FuseArgs f, const int32_t n, const int32_t far_lo, const int32_t far_hi) {
    const bool ended = a.done && *a.done;
    const bool owed = XU && *f.xpend == f.it - 1;
    if (ended && !owed) return;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int XS = T * KR_TILE + 2 * n + 4;                                  // staged elements (even)
    double* xs = reinterpret_cast<double*>(smem);                            // p_old's window, overwritten in place by p_new's
    uint2* meta = reinterpret_cast<uint2*>(xs + XS);
    double* pval = reinterpret_cast<double*>(meta + a.npat);
    double* red = reinterpret_cast<double*>(smem + a.pat_red_off);
    const int t = threadIdx.x;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int G = a.group;
    const int qr = (((slot / G) * 8 + xcd) * G + slot % G) * T;
    if (qr >= a.ntiles) return;
    const int nt = min(T, a.ntiles - qr);
    const int q0 = qr;
    const int32_t r0 = q0 * KR_TILE;
    const double al = *f.alpha, be = *f.beta;
    if (ended) {                                                             // (uniform) only x += alpha p_old is still owed, on the run's own rows
#pragma unroll
        for (int k = 0; k < T; ++k) {
            if (k >= nt) break;
            const int32_t row = r0 + k * KR_TILE + 2 * t;
            const d2 pp = ld2(f.p_old, row), xx = ld2(f.xvec, row);
            st2(f.xvec, row, xx.a + al * pp.a, xx.b + al * pp.b);
        }
        return;
    }
    unsigned ids[T];
#pragma unroll
    // (ids and x are read once: nontemporal, not in the way of z / p_old in L2 -- alternating processes of two builds, 4 rounds: 553 against 552 it/s, no
    // measurable difference beside the 2 % a process's allocation makes; kept because it is the right hint)
    for (int k = 0; k < T; ++k) ids[k] = k < nt ? __builtin_nontemporal_load(reinterpret_cast<const unsigned*>(a.pid + (size_t)(q0 + k) * KR_TILE + 2 * t)) : 0u;
    constexpr int NP = (T * KR_TILE + 2 * NMAX + 4 + 2 * KR_T - 1) / (2 * KR_T);   // pairs per lane at most (n <= NMAX)
    const int npairs = XS / 2;
    const int32_t e0 = r0 - n - 2;                                           // element staged at xs[0] (even)
    const int32_t xsafe = (int32_t)a.xsafe;
    const bool inside = e0 >= 0 && e0 + XS <= xsafe && (int64_t)r0 + far_lo >= 0 && (int64_t)r0 + T * KR_TILE + far_hi <= (int64_t)xsafe;
    const int own_lo = (n + 2) / 2, own_hi = own_lo + nt * (KR_TILE / 2);   // window pairs that are rows of this run
    // ---- the window: p_old straight into LDS (LDS-DMA, lane t's pairs t + 256 i land where lane t reads them back), z into registers
    const int wbase = __builtin_amdgcn_readfirstlane(t & ~63);
    v2d zz[NP], xo[XU ? NP : 1];
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const int pi = t + i * KR_T;
        zz[i].x = 0.0; zz[i].y = 0.0; if constexpr (XU) xo[i] = zz[i];
#ifdef KR_TUNING
        if (pi < npairs && (!(a.abl & 2) || (pi >= own_lo && pi < own_hi))) {   // abl 2 (timing only): the window's halo is not loaded
#else
        if (pi < npairs) {
#endif
            const int32_t e = inside ? e0 + 2 * pi : min(max(e0 + 2 * pi, 0), xsafe);     // outside x: any valid pair (those operands are absent entries)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(f.p_old + e),
                                             (__attribute__((address_space(3))) void*)(xs + 2 * (i * KR_T + wbase)), 16, 0, 0);
            zz[i] = *reinterpret_cast<const v2d*>(f.z + e);
            if constexpr (XU) { if (owed && pi >= own_lo && pi < own_hi) xo[i] = __builtin_nontemporal_load(reinterpret_cast<const v2d*>(f.xvec + e)); }     // (own rows are never clamped; read once)
        }
    }
    for (int i = t; i < a.npat; i += KR_T) meta[i] = reinterpret_cast<const uint2*>(a.pmeta)[i];
    for (int i = t; i < a.ntab; i += KR_T) pval[i] = a.pval[i];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                       // this wave's share of the window has landed (the compiler does not count LDS-DMA requests)
    // p_new = z + beta p_old (cg.rs:274-276 / pcg.rs:215-217, un-fused multiply and add), in place; own rows also to memory, with their x update
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const int pi = t + i * KR_T;
        if (pi < npairs) {
            const v2d po = *reinterpret_cast<const v2d*>(xs + 2 * pi);
            v2d pn; pn.x = zz[i].x + be * po.x; pn.y = zz[i].y + be * po.y;
            *reinterpret_cast<v2d*>(xs + 2 * pi) = pn;
            if (pi >= own_lo && pi < own_hi) {
                const int32_t row = e0 + 2 * pi;
#ifdef KR_TUNING
                if (a.abl & 8) { if (pn.x == 1.23456789e-300) st2(f.p_new, row, pn.x, pn.y); } else      // abl 8 (timing only): no p_new / x stores
#endif
                { st2(f.p_new, row, pn.x, pn.y);
                if constexpr (XU) { if (owed) st2(f.xvec, row, xo[i].x + al * po.x, xo[i].y + al * po.y); } }
            }
        }
    }
    // the far operands of every tile of the run, formed the same way from z and p_old there (requested now: in flight across the barrier)
    v2d lo[T], hi[T];
    {
        v2d zl[T], pl[T], zh[T], ph[T];
#pragma unroll
        for (int k = 0; k < T; ++k) {
            const int64_t row = (int64_t)r0 + k * KR_TILE + 2 * t;
            const int64_t cl = inside ? row + far_lo : min(max(row + far_lo, (int64_t)0), (int64_t)xsafe);
            const int64_t ch = inside ? row + far_hi : min(max(row + far_hi, (int64_t)0), (int64_t)xsafe);
#ifdef KR_TUNING
            if (a.abl & 1) { zl[k].x = (double)cl; zl[k].y = 1.0; pl[k] = zl[k]; zh[k].x = (double)ch; zh[k].y = 2.0; ph[k] = zh[k]; continue; }   // abl 1 (timing only): no far operands
#endif
            zl[k] = *reinterpret_cast<const v2d*>(f.z + cl); pl[k] = *reinterpret_cast<const v2d*>(f.p_old + cl);
            zh[k] = *reinterpret_cast<const v2d*>(f.z + ch); ph[k] = *reinterpret_cast<const v2d*>(f.p_old + ch);
        }
        __syncthreads();                                                     // every wave's p_new pairs are in LDS
#pragma unroll
        for (int k = 0; k < T; ++k) {
            lo[k].x = zl[k].x + be * pl[k].x; lo[k].y = zl[k].y + be * pl[k].y;
            hi[k].x = zh[k].x + be * ph[k].x; hi[k].y = zh[k].y + be * ph[k].y;
        }
    }
    // ---- tile after tile out of LDS (spmv_pattern_stage_kernel's arithmetic)
#pragma unroll
    for (int k = 0; k < T; ++k) {
        if (k >= nt) break;                                                  // (uniform)
        const int32_t row = r0 + k * KR_TILE + 2 * t;
        const bool va = row < a.nrows, vb = row + 1 < a.nrows;
        const uint2 ma = va ? meta[ids[k] & 0xffffu] : make_uint2(0u, 0u), mb = vb ? meta[ids[k] >> 16] : make_uint2(0u, 0u);
        const int j = k * KR_TILE + 2 * t + n + 2;                           // xs[j] = p_new[row]
        const v2d A = *reinterpret_cast<const v2d*>(xs + j - 2), B = *reinterpret_cast<const v2d*>(xs + j), C = *reinterpret_cast<const v2d*>(xs + j + 2);
        const v2d M = *reinterpret_cast<const v2d*>(xs + j - n), P = *reinterpret_cast<const v2d*>(xs + j + n);
        v2d e[7];
        e[0] = lo[k]; e[1] = M; e[2].x = A.y; e[2].y = B.x; e[3] = B; e[4].x = B.y; e[4].y = C.x; e[5] = P; e[6] = hi[k];
        const double* tva = pval + (ma.x & 0xffffu); const double* tvb = pval + (mb.x & 0xffffu);
        const unsigned ka = ma.y, kb = mb.y;
        double s0 = 0.0, s1 = 0.0;
#pragma unroll
        for (int u = 0; u < 7; ++u) {
            const double ta = s0 + tva[u] * e[u].x, tb = s1 + tvb[u] * e[u].y;
            s0 = ((ka >> u) & 1u) ? ta : s0;
            s1 = ((kb >> u) & 1u) ? tb : s1;
        }
#ifdef KR_TUNING
        if (a.abl & 4) { if (s0 == 1.23456789e-300) a.y[row] = s0; } else      // abl 4 (timing only): no y traffic
#endif
        { if (vb) st2(a.y, row, s0, s1);
        else if (va) a.y[row] = s0; }
        if constexpr (NQ > 0) {
            double acc[NQ];
            acc[0] = 0.0;
            if (va) acc[0] = acc[0] + B.x * s0;
            if (vb) acc[0] = acc[0] + B.y * s1;
            if constexpr (NQ > 1) {
                acc[1] = 0.0;
                if (va) acc[1] = acc[1] + s0 * s0;
                if (vb) acc[1] = acc[1] + s1 * s1;
            }
#pragma unroll
            for (int kk = 0; kk < NQ; ++kk) {
                const double wsum = wave_butterfly(acc[kk]);
                if ((t & 63) == 0) red[(k * NQ + kk) * (KR_T / 64) + (t >> 6)] = wsum;
            }
        }
    }
    if constexpr (NQ > 0) {
        __syncthreads();
        if (t < nt * NQ) {
            const int k = t / NQ, kk = t % NQ;
            double sum = red[t * (KR_T / 64)];
#pragma unroll
            for (int w2 = 1; w2 < KR_T / 64; ++w2) sum = sum + red[t * (KR_T / 64) + w2];
            a.partials[kk * a.pstride + q0 + k] = sum;
        }
    }
}

// CSR-DIA (spmv_dia_kernel): operators with a handful of well-filled diagonals -- every stencil on a structured grid, variable
// coefficients included.  The values are stored as one stream per diagonal in natural row order (csr_create.hip: build_dia), an
// absent entry carries a NaN payload no stored value may have, so there are NO row pointers, NO per-entry codes and no
// dependent load: lane t (rows 2t, 2t+1) issues, per diagonal, one 16-byte nontemporal load of its two values and one 16-byte
// load of x[2t + off], x[2t + 1 + off] -- all 2 D loads of a tile in flight at once -- and then folds the present entries in
// diagonal (= stored, ascending column) order from 0.0 with separate mul and add: the reference's loop (sparse.rs:107-113), bit
// for bit.  8 D + 16 bytes per row (7-point: 72; CSR-D8 83, plain CSR 104).  Pair loads of x are clamped at the top of x's
// (padded) allocation -- only an absent entry can point there -- and tiles whose lowest diagonal would start before x[0], as well
// as tiles with halo columns, take the per-element clamped path.
// one batch of DB diagonals starting at d0 (EXACT: the operator has exactly DB diagonals, so every index is a compile-time constant
// and the offsets are plain kernel arguments); FAST: 16-byte pair loads of x, else per-element clamped gathers
template <bool HALO, int DB, bool EXACT, bool FAST>
__device__ __forceinline__ void dia_batch(const SpmvArgs& a, int row, int d0, int nd, double& s0, double& s1) {
    v2d v[DB], xx[DB];
    const double* vrow = a.dia + row;
#pragma unroll
    for (int u = 0; u < DB; ++u) {
        const int d = EXACT ? u : min(d0 + u, nd - 1);           // (a batch's tail re-reads the last diagonal; not used below)
        v[u] = __builtin_nontemporal_load(reinterpret_cast<const v2d*>(vrow + (int64_t)d * a.dia_stride));
        const int64_t c = (int64_t)row + a.dia_off[d];
        if constexpr (FAST) {
            xx[u] = *reinterpret_cast<const v2d*>(a.x + min(c, a.xsafe));
        } else {
            const int32_t c0 = (int32_t)min(max(c, (int64_t)0), (int64_t)a.cmax), c1 = (int32_t)min(max(c + 1, (int64_t)0), (int64_t)a.cmax);
            xx[u].x = gather<HALO>(a, c0); xx[u].y = gather<HALO>(a, c1);
        }
    }
#pragma unroll
    for (int u = 0; u < DB; ++u) {
        if (EXACT || d0 + u < nd) {
            const double ta = s0 + v[u].x * xx[u].x, tb = s1 + v[u].y * xx[u].y;
            s0 = ((unsigned long long)__double_as_longlong(v[u].x) != KR_DIA_ABSENT) ? ta : s0;
            s1 = ((unsigned long long)__double_as_longlong(v[u].y) != KR_DIA_ABSENT) ? tb : s1;
        }
    }
}

template <int NQ, bool HALO, int DB, bool EXACT>
__global__ __launch_bounds__(KR_T) void spmv_dia_kernel(const SpmvArgs a) {
    if (a.done && *a.done) return;
    __shared__ double red[(NQ > 0 ? NQ : 1) * (KR_T / 64)];
    const int t = threadIdx.x;
    const int xcd = blockIdx.x & 7, slot0 = blockIdx.x >> 3, per = gridDim.x >> 3;
    const int nd = a.dia_nd;
    for (int li = slot0; li < a.xcd_chunk; li += per) {
        int ti;
        if (a.swizzle) ti = xcd * a.xcd_chunk + li;
        else ti = ((li / a.group) * 8 + xcd) * a.group + (li % a.group);
        if (ti >= a.ntiles) { if (a.swizzle) break; else continue; }
        const int q = a.tiles ? a.tiles[ti] : ti;
        if (q < 0) continue;                                    // an empty slot of the slab order
        const int r0 = q * KR_TILE;
        const int r1 = min(r0 + KR_TILE, a.nrows);
        const int row = r0 + 2 * t;
        double s0 = 0.0, s1 = 0.0;
        if (!HALO && r0 + a.dia_min >= 0) {                            // uniform over the workgroup
            for (int d0 = 0; d0 < (EXACT ? 1 : nd); d0 += DB) dia_batch<HALO, DB, EXACT, true>(a, row, d0, nd, s0, s1);
        } else {
            for (int d0 = 0; d0 < (EXACT ? 1 : nd); d0 += DB) dia_batch<HALO, DB, EXACT, false>(a, row, d0, nd, s0, s1);
        }
        if (row + 1 < r1) st2(a.y, row, s0, s1);
        else if (row < r1) a.y[row] = s0;
        if constexpr (NQ > 0) {
            double acc[NQ];
            const d2 d = ld2(a.dvec, row);
            acc[0] = 0.0;
            if (row < r1) acc[0] = acc[0] + d.a * s0;
            if (row + 1 < r1) acc[0] = acc[0] + d.b * s1;
            if constexpr (NQ > 1) {
                acc[1] = 0.0;
                if (row < r1) acc[1] = acc[1] + s0 * s0;
                if (row + 1 < r1) acc[1] = acc[1] + s1 * s1;
            }
            block_reduce<NQ, KR_T / 64>(acc, red);
            if (t == 0) {
#pragma unroll
                for (int k = 0; k < NQ; ++k) a.partials[k * a.pstride + q] = acc[k];
            }
        }
    }
}

__global__ void pack_kernel(const double* x, const int32_t* idx, double* out, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = x[idx[i]];
}

// tuning knobs (read per launch so that one process can A/B them)
static int spmv_blocks_per_cu() { return env_int("KRYST_SPMV_BLOCKS_PER_CU", 0); }

// Which kernel a launch takes is decided from the forms the operator has and the KRYST_SPMV_* settings (read per launch).
// Runs per XCD group of the staged-window kernels (spmv_pattern_stage_kernel / _fuse_kernel: groups of G consecutive runs round-robin over the 8
// XCDs): groups of 4 once the launch is well beyond one wave of workgroups (in-process A/B inside CG: 512^3 +1.2 %, 256^3 +-0, 192^3 +4 %, but
// 128^3 -19 %; 8: +1.3 / -4.5 % at 512^3 / 256^3; 32: -2 / -21 %).  The FUSED kernel reads its far operands (rows +- one plane) from TWO vectors:
// when a grid plane (far_hi rows) is a whole number of groups per XCD -- G = plane / (8 T tiles) -- the same (i, j) strip of every plane lands on
// the same XCD, whose L2 then streams those rows itself: 512^3 fused CG, G = 16 against 4: 613 against 596 it/s (three interleaved rounds,
// tools/cg_fuse_knobs.py).  (Walking SUB-strips of R runs plane by plane -- resident runs then neighbours in k -- was measured and loses: R = 1 / 2 / 4 /
// 8 / 16: 563 / 578 / 590 / 603 / 616 it/s; a run that jumps a plane every few runs costs DRAM page locality more than the far operands' L2 hits give.)  The plain staged kernel does not care at 512^3 (0.500-0.508 ms for G = 1 .. 16, tools/stage_group_ab.py) and loses
// with large groups at 384^3 (0.26 against 0.21 ms), so it keeps 4.
static int stage_group_default(kryst_csr_t a, int64_t nruns, int T, bool fused) {
    if (nruns < 2048) return 1;
    const int64_t plane = a->pat_far_uniform ? (int64_t)a->pat_far_hi : 0, per = (int64_t)KR_TILE * 8 * T;
    if (fused && plane > 0 && plane % per == 0 && plane / per >= 2 && plane / per <= 64) return (int)(plane / per);
    return 4;
}
static bool takes_pattern_path(kryst_csr_t a, bool halo) {
    return a->d_pid && env_int("KRYST_SPMV_COMPRESS", 3) >= 3 && a->xlen + (halo ? a->plan.total_recv : 0) < (1ll << 28) && a->nrows < (1ll << 28);
}
// slab order (csr_create.hip: build_tile_order) for a whole local operator.  KRYST_SPMV_ORDER: 1 (default) where it was measured
// to pay -- the CSR-DIA kernel, and the plain kernel once two planes of x outgrow an XCD's L2 (in-process A/B, tools/spmv_ab.py,
// profiles/r03/slab_order/: plain 512^3 +1.3-2.2 %, 448^3 -1.4 %, 384^3 -1.1 %, 256^3 -2.2 %; CSR-DIA 512^3 / 256^3 +0.4 / +1.9 %;
// CSR-D8 -2.4 / -3.6 %, CSR-D16 -0.7 / -2.9 %, CSR-P16 -4.8 / -4.6 % -- although the x re-reads go away in every form: 17.3 -> 14.3 GB
// per launch through the fabric for plain CSR at 512^3, 11.8 -> 9.7 GB for CSR-DIA); 2 every kernel; 0 never
static bool uses_tile_order(kryst_csr_t a) {
    if (!a->d_tile_order || a->dist) return false;
    const int order_mode = env_int("KRYST_SPMV_ORDER", 1);
    if (order_mode <= 0) return false;
    if (order_mode >= 2) return true;
    const int form_level = env_int("KRYST_SPMV_COMPRESS", 3);
    if (takes_pattern_path(a, false) || (a->d_code16 && form_level >= 2)) return false;
    if (a->d_dia && form_level >= 1 && env_int("KRYST_SPMV_DIA", 1) != 0) return true;
    const bool plain_path = !(a->d_code && form_level != 0) && env_int("KRYST_SPMV_KERNEL", 2) != 3;
    return plain_path && a->order_plane >= 262144;
}

template <bool HALO>
static int32_t launch_tiles(kryst_csr_t a, const double* x, double* y, int nq, const double* dvec, const int* done,
                            const int32_t* tiles, int64_t ntiles) {
    if (ntiles == 0) return KRYST_OK;
    kryst_ctx_t ctx = a->ctx;
    SpmvArgs args;
    args.row_ptr = a->d_row_ptr; args.col = a->d_col; args.val = a->d_val;
    args.x = x; args.halo = a->plan.d_halo; args.nloc = (int32_t)a->nrows;
    const bool pattern_path = takes_pattern_path(a, HALO);
    const bool ordered = !tiles && !HALO && ntiles == a->ntiles && uses_tile_order(a);
    if (ordered) {
        tiles = pattern_path ? a->d_tile_order + a->order_slots1 : a->d_tile_order;
        ntiles = pattern_path ? a->order_slots8 : a->order_slots1;
    }
    args.y = y; args.tiles = tiles; args.ntiles = (int32_t)ntiles; args.nrows = (int32_t)a->nrows;
    args.dvec = dvec; args.partials = ctx->d_partials; args.pstride = ctx->partials_cap; args.done = done;
    args.swizzle = env_int("KRYST_SPMV_SWIZZLE", 0);
    // the plain kernel beyond the Infinity Cache: nontemporal matrix streams in windows that start on whole memory lines (in-process
    // A/B, tools/spmv_ab.py, profiles/r03/slab_order/: 384^3 +0.8 %, 448^3 +1.6-2.0 %, 512^3 +1.1-2.1 %; nontemporal loads alone lose 1 %)
    const bool beyond_cache = a->nrows * 8 > (256ll << 20);
    args.amask = env_int("KRYST_SPMV_ALIGN", beyond_cache ? 1 : 0) ? 31 : 1;
#ifdef KR_TUNING
    args.abl = env_int("KRYST_SPMV_ABL", 0);
#endif
    args.group = ordered ? 1 : std::max(1, env_int("KRYST_SPMV_GROUP", 1));
    if (ordered) args.swizzle = 0;
    int64_t chunk = (ntiles + 7) / 8;                       // tiles per XCD
    if (!args.swizzle) chunk = (chunk + args.group - 1) / args.group * args.group;   // a whole number of runs of `group` tiles
    args.xcd_chunk = (int32_t)chunk;
    int64_t per = chunk;                                    // one tile per workgroup by default
    const int bpc = spmv_blocks_per_cu();
    if (bpc > 0) per = std::min<int64_t>(chunk, std::max<int64_t>(1, (int64_t)ctx->num_cu * bpc / 8));
    const dim3 grid((unsigned)(per * 8)), block(KR_T);
    // production path: wave-independent kernel; pair slots per lane sized to the matrix's average slice length
    // (a wave streams the nnz of 128 rows: <= 256 nnz -> 2 slots, <= 512 -> 4, else 7 = a 7-point stencil's 896)
    const int kern = env_int("KRYST_SPMV_KERNEL", 2);          // 2: products-in-LDS wave kernel; 3: rows kernel (+ CSR-D8)
    // 0 plain CSR, 1 CSR-D8 (offset codes), 2 CSR-D16 (offset + value codes), 3 CSR-P16 (row patterns); each level
    // falls back to the next lower one the operator qualifies for
    const int comp_level = env_int("KRYST_SPMV_COMPRESS", 3);
    const bool comp = a->d_code && comp_level != 0;
    args.code = comp ? a->d_code : nullptr; args.dict = comp ? a->d_dict : nullptr;
    args.code16 = a->d_code16; args.vdict = a->d_vdict;
    args.pid = a->d_pid; args.pmeta = a->d_pmeta; args.poff = a->d_poff; args.pval = a->d_pval; args.npat = a->npat; args.ntab = a->ntab;
    if (pattern_path) {   // (32-bit byte offsets: xlen < 2^28)
        // a workgroup loads the tables once and walks one run of 8 consecutive tiles (next tile's ids prefetched); runs go
        // round-robin over the XCDs.  Measured at 512^3 (round-2 sweeps and --pmc passes, profiles/r02/): 0.70 ms and 1.6 GB of reads
        // per launch, against 0.79 ms and 4.1 GB for a strided persistent grid whose fast workgroups run ahead.
        // the near operands staged in LDS, runs of T = 4 tiles per workgroup (natural tile order, no halo columns)
        const bool whole = !tiles, interior = tiles && tiles == a->d_tiles_interior && a->interior_first >= 0 && ntiles == a->n_interior;
        if (!HALO && (whole || interior) && a->pat_stage_n > 0 && a->npat <= 512 && a->ntab <= 512 && env_int("KRYST_SPMV_STAGE", 1) != 0 &&
            a->xlen + 2 * KR_TILE < (1ll << 31)) {
            const int32_t tile0 = interior ? (int32_t)a->interior_first : 0;
            args.tiles = nullptr;
            // runs of 4 tiles for large vectors, of 2 below half a gigabyte (63 registers, 8 waves per SIMD: 512^3 2 % slower, 256^3 3.5 % faster)
            const int T = env_int("KRYST_SPMV_STAGE_T", a->nrows * 8 > (512ll << 20) ? 4 : 2) <= 2 ? 2 : 4;     // (in CG: 128^3 +1.2 %, 192^3 +2 %, 256^3 +0.6 %, 320^3 +1.3 % with 2)
            const int32_t n_ = a->pat_stage_n;
            const size_t xs_bytes = sizeof(double) * (size_t)(T * KR_TILE + 2 * n_ + 4);
            const size_t tab = xs_bytes + (size_t)a->npat * 8 + (size_t)a->ntab * 12;
            args.pat_red_off = (int32_t)((tab + 15) & ~(size_t)15);
            const size_t lds_s = (size_t)args.pat_red_off + sizeof(double) * (size_t)T * (size_t)(nq > 0 ? nq : 1) * (KR_T / 64);
            args.xsafe = (a->xlen + KR_TILE - 1) / KR_TILE * KR_TILE + KR_TILE - 2;
            const int64_t nruns = (ntiles + T - 1) / T;
            args.group = std::max(1, env_int("KRYST_SPMV_STAGE_GROUP", stage_group_default(a, nruns, T, false)));
            const int64_t per_xcd = ((nruns + 7) / 8 + args.group - 1) / args.group * args.group;      // slots per XCD: whole groups
            const dim3 sgrid((unsigned)(per_xcd * 8));
            const bool center = nq > 0 && dvec == x;
            const bool ufar = (a->pat_far_uniform || (interior && a->pat_far_interior)) && env_int("KRYST_SPMV_STAGE_UFAR", 1) != 0;
#define KR_STG_T(NQ_, C_, T_) do { if (ufar) hipLaunchKernelGGL((spmv_pattern_stage_kernel<NQ_, T_, C_, true>), sgrid, block, lds_s, ctx->s_main, args, n_, a->pat_far_lo, a->pat_far_hi, tile0); \
                                   else hipLaunchKernelGGL((spmv_pattern_stage_kernel<NQ_, T_, C_, false>), sgrid, block, lds_s, ctx->s_main, args, n_, 0, 0, tile0); } while (0)
#define KR_STG(NQ_, C_) do { if (T == 2) KR_STG_T(NQ_, C_, 2); else KR_STG_T(NQ_, C_, 4); } while (0)
            switch (nq) {
                case 0: KR_STG(0, false); break;
                case 1: if (center) KR_STG(1, true); else KR_STG(1, false); break;
                case 2: if (center) KR_STG(2, true); else KR_STG(2, false); break;
                default: set_error("spmv: nq=%d", nq); return KRYST_ERR_ARG;
            }
#undef KR_STG
#undef KR_STG_T
            KR_HIP(hipGetLastError());
            return KRYST_OK;
        }
        args.group = ordered ? 8 : std::max(1, env_int("KRYST_SPMV_GROUP", 8));
        const int64_t pchunk = (chunk + args.group - 1) / args.group * args.group;      // an XCD's share is a whole number of runs
        args.xcd_chunk = (int32_t)pchunk;
        args.tpw = std::max(1, env_int("KRYST_SPMV_PATTERN_TPW", 8));
        const int64_t pper = (pchunk + args.tpw - 1) / args.tpw;
        const dim3 pgrid((unsigned)(pper * 8));
        const int U = a->pat_unroll;
        const size_t tab = (size_t)(a->ntab + U + 1) * 12 + (size_t)a->npat * 8;
        args.pat_red_off = (int32_t)((tab + 15) & ~(size_t)15);
        args.nloc8 = (int32_t)(a->nrows * 8);
        const size_t lds = (size_t)args.pat_red_off + sizeof(double) * (size_t)(nq > 0 ? nq : 1) * (KR_T / 64);
#define KR_PAT(NQ_, U_, S_) hipLaunchKernelGGL((spmv_pattern_kernel<NQ_, HALO, U_, S_>), pgrid, block, lds, ctx->s_main, args)
#define KR_PAT_C(NQ_) hipLaunchKernelGGL((spmv_pattern_kernel<NQ_, HALO, 7, true, 3, true>), pgrid, block, lds, ctx->s_main, args)
#define KR_PAT_M(NQ_) hipLaunchKernelGGL((spmv_pattern_kernel<NQ_, HALO, 7, true, -1, true>), pgrid, block, lds, ctx->s_main, args)
        const bool gen3 = !HALO && a->pat_diag3 && env_int("KRYST_SPMV_REUSE_DIAG", 1);      // generator-made operator, interior tiles
        const bool reuse_diag = gen3 && nq > 0 && dvec == x;
#define KR_PAT_BY(NQ_) do { if (U == 7 && reuse_diag && NQ_ > 0) KR_PAT_C(NQ_); else if (U == 7 && gen3) KR_PAT_M(NQ_); else if (U == 7) KR_PAT(NQ_, 7, true); \
                            else if (a->pat_single) KR_PAT(NQ_, 8, true); else KR_PAT(NQ_, 8, false); } while (0)
        switch (nq) {
            case 0: KR_PAT_BY(0); break;
            case 1: KR_PAT_BY(1); break;
            case 2: KR_PAT_BY(2); break;
            default: set_error("spmv: nq=%d", nq); return KRYST_ERR_ARG;
        }
#undef KR_PAT_BY
#undef KR_PAT_C
#undef KR_PAT_M
#undef KR_PAT
        KR_HIP(hipGetLastError());
        return KRYST_OK;
    }
    if (a->d_code16 && comp_level >= 2) {
        switch (nq) {
            case 0: hipLaunchKernelGGL((spmv_dict_kernel<0, HALO>), grid, block, 0, ctx->s_main, args); break;
            case 1: hipLaunchKernelGGL((spmv_dict_kernel<1, HALO>), grid, block, 0, ctx->s_main, args); break;
            case 2: hipLaunchKernelGGL((spmv_dict_kernel<2, HALO>), grid, block, 0, ctx->s_main, args); break;
            default: set_error("spmv: nq=%d", nq); return KRYST_ERR_ARG;
        }
        KR_HIP(hipGetLastError());
        return KRYST_OK;
    }
    if (a->d_dia && comp_level >= 1 && env_int("KRYST_SPMV_DIA", 1) != 0) {
        args.dia = a->d_dia; args.dia_stride = a->dia_stride; args.dia_nd = a->dia_nd; args.dia_min = a->dia_min;
        for (int d = 0; d < KR_DIA_MAX; ++d) args.dia_off[d] = a->dia_off[d];
        args.xsafe = (a->xlen + KR_TILE - 1) / KR_TILE * KR_TILE + KR_TILE - 2;      // vectors are allocated padded to a tile multiple + one tile (blas1.hip)
        args.cmax = (int32_t)(HALO ? a->nrows + a->plan.total_recv - 1 : a->xlen - 1);
        const int nd = a->dia_nd;
#define KR_DIA(NQ_, DB_, EX_) hipLaunchKernelGGL((spmv_dia_kernel<NQ_, HALO, DB_, EX_>), grid, block, 0, ctx->s_main, args)
#define KR_DIA_BY(NQ_) do { if (nd == 3) KR_DIA(NQ_, 3, true); else if (nd == 5) KR_DIA(NQ_, 5, true); else if (nd == 7) KR_DIA(NQ_, 7, true); \
                            else if (nd == 9) KR_DIA(NQ_, 9, true); else if (nd < 7) KR_DIA(NQ_, 4, false); else KR_DIA(NQ_, 8, false); } while (0)
        switch (nq) {
            case 0: KR_DIA_BY(0); break;
            case 1: KR_DIA_BY(1); break;
            case 2: KR_DIA_BY(2); break;
            default: set_error("spmv: nq=%d", nq); return KRYST_ERR_ARG;
        }
#undef KR_DIA_BY
#undef KR_DIA
        KR_HIP(hipGetLastError());
        return KRYST_OK;
    }
    const int slots_env = env_int("KRYST_SPMV_SLOTS", 0);
    // measured (tools/tune_spmv.py): with vectors beyond the 256 MiB Infinity Cache the smaller window (more resident
    // waves) wins, below it the one-window-per-slice form does
    int slots_sel = a->slots;
    if (comp && a->slots > 4 && a->nrows * 8 > (256ll << 20)) slots_sel = 4;
    if (slots_env > 0) slots_sel = slots_env;
    if (kern == 3 || comp) {
#define KR_ROWS(NQ_, SL_, C_) hipLaunchKernelGGL((spmv_rows_kernel<NQ_, HALO, SL_, C_>), grid, block, 0, ctx->s_main, args)
#define KR_ROWS_BY(NQ_)                                                                 \
        do {                                                                            \
            if (comp) { if (slots_sel <= 4) KR_ROWS(NQ_, 4, true); else KR_ROWS(NQ_, 7, true); }      \
            else { if (slots_sel <= 4) KR_ROWS(NQ_, 4, false); else KR_ROWS(NQ_, 7, false); }         \
        } while (0)
        switch (nq) {
            case 0: KR_ROWS_BY(0); break;
            case 1: KR_ROWS_BY(1); break;
            case 2: KR_ROWS_BY(2); break;
            default: set_error("spmv: nq=%d", nq); return KRYST_ERR_ARG;
        }
#undef KR_ROWS_BY
#undef KR_ROWS
        KR_HIP(hipGetLastError());
        return KRYST_OK;
    }
    const bool nt = env_int("KRYST_SPMV_NT", beyond_cache ? 1 : 0) != 0;
    // window size of the plain kernel: one window per 128-row slice (7 pair slots for a 7-point stencil's 896 entries) while the
    // vectors fit the 256 MiB Infinity Cache, 4 slots (74 registers, 6 waves per SIMD instead of 4) beyond it -- measured
    // (profiles/r02/plain_csr_study/README.md): 256^3 0.298 / 0.309 ms with 7 / 4 slots, 512^3 2.84 / 2.62 ms
    int wslots = a->slots;
    if (wslots > 4 && a->nrows * 8 > (256ll << 20)) wslots = 4;
    if (slots_env > 0) wslots = slots_env;
#define KR_SPMV_LAUNCH(NQ_, SL_) do { if (nt) hipLaunchKernelGGL((spmv_wave_kernel<NQ_, HALO, SL_, true>), grid, block, 0, ctx->s_main, args); \
                                      else hipLaunchKernelGGL((spmv_wave_kernel<NQ_, HALO, SL_, false>), grid, block, 0, ctx->s_main, args); } while (0)
#define KR_SPMV_BY_SLOTS(NQ_)                                   \
    do {                                                        \
        if (wslots <= 2) KR_SPMV_LAUNCH(NQ_, 2);                \
        else if (wslots <= 4) KR_SPMV_LAUNCH(NQ_, 4);           \
        else KR_SPMV_LAUNCH(NQ_, 7);                            \
    } while (0)
    switch (nq) {
        case 0: KR_SPMV_BY_SLOTS(0); break;
        case 1: KR_SPMV_BY_SLOTS(1); break;
        case 2: KR_SPMV_BY_SLOTS(2); break;
        default: set_error("spmv: nq=%d", nq); return KRYST_ERR_ARG;
    }
#undef KR_SPMV_BY_SLOTS
#undef KR_SPMV_LAUNCH
    KR_HIP(hipGetLastError());
    return KRYST_OK;
}

void halo_send_tiles(kryst_csr_t a, std::vector<std::pair<int64_t, int64_t>>& ranges) {
    ranges.clear();
    if (!a->dist || !a->send_contiguous) return;
    const HaloPlan& pl = a->plan;
    for (size_t p = 0; p < pl.send_counts.size(); ++p)
        if (pl.send_counts[p] > 0) ranges.emplace_back(pl.send_off[p] / KR_TILE, (pl.send_off[p] + pl.send_counts[p] + KR_TILE - 1) / KR_TILE);
    std::sort(ranges.begin(), ranges.end());
    std::vector<std::pair<int64_t, int64_t>> merged;
    for (const auto& r : ranges) {
        if (!merged.empty() && r.first <= merged.back().second) merged.back().second = std::max(merged.back().second, r.second);
        else merged.push_back(r);
    }
    ranges.swap(merged);
}

// ---- halo exchange by direct peer stores (dist.h: HaloPeer)
// push: workgroup (bx, seg) stores its share of segment `seg` into the neighbour's landing buffer; the workgroup that finishes LAST (all
// data stores of the launch acknowledged: write-through system-scope stores + s_waitcnt vmcnt(0) in front of the ticket) stamps the epoch
// into every neighbour's cell -- the protocol of fold_ipc_logic_kernel (solver_common.h).
__global__ __launch_bounds__(256) void halo_push_kernel(const double* __restrict__ x, const int32_t* __restrict__ send_idx, const HaloPushSeg* __restrict__ segs,
                                                        int nsegs, int parity, unsigned long long epoch, unsigned int* ticket) {
    const HaloPushSeg sg = segs[blockIdx.y];
    double* dst = sg.dst + (size_t)parity * sg.dst_stride;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < sg.count; i += (int64_t)gridDim.x * 256) {
        const double v = send_idx ? x[send_idx[sg.src + i]] : x[sg.src + i];
        __hip_atomic_store(dst + i, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __shared__ int last;
    __syncthreads();                                                  // every wave of the workgroup has had its stores acknowledged
    if (threadIdx.x == 0) {
        const unsigned int t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        last = t == gridDim.x * gridDim.y - 1 ? 1 : 0;
    }
    __syncthreads();
    if (!last) return;
    if (threadIdx.x < nsegs) __hip_atomic_store(segs[threadIdx.x].stamp, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (threadIdx.x == 0) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // ready for the next (stream-ordered) push
}
// pull: wait for the sender's stamp, then landing[parity] -> d_halo (uncached reads: the landing buffer is written by another agent).
// A stamp that never comes (budget) poisons the halo with NaNs and raises the context's error word instead of hanging the GPU.
__global__ __launch_bounds__(256) void halo_pull_kernel(const double* __restrict__ landing, double* __restrict__ halo, const HaloPullSeg* __restrict__ segs,
                                                        unsigned long long epoch, int budget, unsigned int* err) {
    const HaloPullSeg sg = segs[blockIdx.y];
    __shared__ int ok;
    if (threadIdx.x == 0) {
        // (a pull that has given up once on this context stays given up until the host has read the error word: the exchanges a solver has
        // already enqueued behind it must not burn the budget again, one after the other)
        int b = __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u ? 1 : budget;
        unsigned long long seen = __hip_atomic_load(sg.stamp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        while (seen < epoch && --b > 0) { __builtin_amdgcn_s_sleep(2); seen = __hip_atomic_load(sg.stamp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
        ok = seen >= epoch ? 1 : 0;
        if (!ok) __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    const bool good = ok != 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < sg.count; i += (int64_t)gridDim.x * 256)
        halo[sg.off + i] = good ? __hip_atomic_load(landing + sg.off + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : __longlong_as_double(0x7FF8000000000000ll);
}

// the landing buffer's first state (data and stamps zero), written with the stores its later writers use
__global__ void halo_landing_init_kernel(double* landing, int64_t count) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) __hip_atomic_store(landing + i, 0.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

void halo_peer_destroy(kryst_csr_t a) {
    HaloPeer& hp = a->plan.peer;
    for (void* q : hp.opened) ipc_close_shared(q);
    hp.opened.clear();
    // A landing buffer that a rank THREAD of this process stores into by raw pointer (no hipIpc mapping keeps the memory alive for it) outlives
    // the operator: a slower neighbour's last early push (solvers.hip: launch_direction) may still be queued when this rank destroys its
    // operator.  It is freed with the context (ctx.cpp), after the streams of every rank of a rehearsal have drained.
    if (hp.landing && hp.sibling) { a->ctx->deferred_free.push_back(hp.landing); hp.landing = nullptr; }
    (void)hipFree(hp.landing); (void)hipFree(hp.d_push); (void)hipFree(hp.d_pull); (void)hipFree(hp.d_ticket);
    hp.landing = nullptr; hp.d_push = nullptr; hp.d_pull = nullptr; hp.d_ticket = nullptr; hp.on = false;
    (void)hipGetLastError();
}

int32_t halo_peer_setup(kryst_csr_t a) {
    kryst_ctx_t ctx = a->ctx;
    HaloPlan& pl = a->plan;
    HaloPeer& hp = pl.peer;
    if (hp.on) return KRYST_OK;
    if (hp.landing && hp.failed) { set_error("halo by peer stores: the test exchange failed on this operator before; the RCCL exchange stays in use"); return KRYST_UNSUPPORTED; }
    if (hp.landing) { hp.on = true; return KRYST_OK; }               // set up before and switched off: still mapped everywhere, epochs counted alike
    KR_ARG(a->dist && ctx->comm, "halo_peer_setup: not a distributed operator");
    const int P = ctx->nranks, me = ctx->rank;
    KR_ARG(P <= 257, "halo by peer stores: at most 256 neighbours (one stamping lane each)");      // (the same answer on every rank: P is)
    KR_HIP(hipStreamSynchronize(ctx->s_comm));
    KR_HIP(hipStreamSynchronize(ctx->s_main));
    // local: landing buffer (zeroed: stamps start at epoch 0); a rank that fails here, or whose neighbour relations are not mutual, still
    // takes part in the collectives below with a null buffer, which then fail on every rank alike
    bool mutual = true;
    for (int p = 0; p < P; ++p) mutual = mutual && ((pl.send_counts[p] > 0) == (pl.recv_counts[p] > 0));
    hp.stride = std::max<int64_t>(2, (pl.total_recv + 1) & ~(int64_t)1);
    const size_t bytes = sizeof(double) * (size_t)(2 * hp.stride + 2 * P);
    // (everything this rank allocates is allocated BEFORE the collectives: a rank that fails alone afterwards would leave the others switched on)
    if (!mutual || hipMalloc(&hp.d_push, sizeof(HaloPushSeg) * ((size_t)P + 1)) != hipSuccess || hipMalloc(&hp.d_pull, sizeof(HaloPullSeg) * ((size_t)P + 1)) != hipSuccess ||
        hipMalloc(&hp.d_ticket, 64) != hipSuccess || hipMemsetAsync(hp.d_ticket, 0, 64, ctx->s_main) != hipSuccess ||
        hipExtMallocWithFlags((void**)&hp.landing, std::max<size_t>(bytes, (size_t)2 << 20), hipDeviceMallocFinegrained) != hipSuccess) {   // (an allocation of its own: dist.cpp, ipc_reduce_setup)
        (void)hipGetLastError();
        (void)hipFree(hp.landing); hp.landing = nullptr;
    }
    if (hp.landing) {
        const int64_t cnt = (int64_t)(bytes / sizeof(double));
        hipLaunchKernelGGL(halo_landing_init_kernel, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, ctx->s_main, hp.landing, cnt);
        if (hipGetLastError() != hipSuccess || hipStreamSynchronize(ctx->s_main) != hipSuccess) { (void)hipGetLastError(); (void)hipFree(hp.landing); hp.landing = nullptr; }
    }
    std::vector<void*> peers;
    // (a pull kernel waits for a push its neighbour's host thread enqueues LATER -- at its next exchange: rank threads that share one device are refused)
    int32_t rc = ipc_map_peers(ctx, hp.landing, peers, hp.opened, false, &hp.sibling);
    if (rc != KRYST_OK) { halo_peer_destroy(a); return rc; }
    // where my rows land in each neighbour's buffer: every rank's {stride, recv_off[0..P)} in one all-gather
    std::vector<int64_t> mine((size_t)P + 1), all((size_t)(P + 1) * P, 0);
    mine[0] = hp.stride;
    for (int p = 0; p < P; ++p) mine[(size_t)p + 1] = pl.recv_off[p];
    int64_t *d_s = nullptr, *d_r = nullptr;
    if (hipMalloc(&d_s, sizeof(int64_t) * (P + 1)) != hipSuccess || hipMalloc(&d_r, sizeof(int64_t) * (size_t)(P + 1) * P) != hipSuccess) rc = KRYST_ERR_HIP;
    if (rc == KRYST_OK && (hipMemcpyAsync(d_s, mine.data(), sizeof(int64_t) * (P + 1), hipMemcpyHostToDevice, ctx->s_main) != hipSuccess || hipStreamSynchronize(ctx->s_main) != hipSuccess)) rc = KRYST_ERR_HIP;
    if (rc == KRYST_OK) rc = comm_all_gather_i64(ctx, d_s, d_r, P + 1, ctx->s_main);
    if (rc == KRYST_OK && (hipMemcpyAsync(all.data(), d_r, sizeof(int64_t) * all.size(), hipMemcpyDeviceToHost, ctx->s_main) != hipSuccess || hipStreamSynchronize(ctx->s_main) != hipSuccess)) rc = KRYST_ERR_HIP;
    (void)hipFree(d_s); (void)hipFree(d_r);
    if (rc != KRYST_OK) { halo_peer_destroy(a); return rc; }         // (a failing collective fails on every rank)
    std::vector<HaloPushSeg> push; std::vector<HaloPullSeg> pull;
    hp.push_max = hp.pull_max = 0;
    // (send_off: for contiguous lists the first local row of the run, otherwise the list's first entry in d_send_idx -- csr_create.hip)
    for (int p = 0; p < P; ++p) {
        if (p != me && pl.send_counts[p] > 0) {
            const int64_t pstride = all[(size_t)(P + 1) * p], poff = all[(size_t)(P + 1) * p + 1 + me];
            double* base = static_cast<double*>(peers[p]);
            push.push_back(HaloPushSeg{base + poff, reinterpret_cast<unsigned long long*>(base + 2 * pstride + 2 * me), pstride, pl.send_off[p], pl.send_counts[p]});
            hp.push_max = std::max(hp.push_max, pl.send_counts[p]);
        }
        if (p != me && pl.recv_counts[p] > 0) {
            pull.push_back(HaloPullSeg{reinterpret_cast<const unsigned long long*>(hp.landing + 2 * hp.stride + 2 * p), pl.recv_off[p], pl.recv_counts[p]});
            hp.pull_max = std::max(hp.pull_max, pl.recv_counts[p]);
        }
    }
    hp.npush = (int)push.size(); hp.npull = (int)pull.size();
    if (env_int("KRYST_HALO_DEBUG", 0)) {
        fprintf(stderr, "[kryst halo rank %d] landing %p stride %lld total_recv %lld\n", me, (void*)hp.landing, (long long)hp.stride, (long long)pl.total_recv);
        for (int p = 0; p < P; ++p) fprintf(stderr, "[kryst halo rank %d]   peer %d -> %p\n", me, p, peers[p]);
        for (auto& q : push) fprintf(stderr, "[kryst halo rank %d]   push dst %p stamp %p dst_stride %lld src %lld count %lld\n", me, (void*)q.dst, (void*)q.stamp, (long long)q.dst_stride, (long long)q.src, (long long)q.count);
        for (auto& q : pull) fprintf(stderr, "[kryst halo rank %d]   pull stamp %p off %lld count %lld\n", me, (const void*)q.stamp, (long long)q.off, (long long)q.count);
    }
    if (!push.empty()) KR_HIP(hipMemcpyAsync(hp.d_push, push.data(), sizeof(HaloPushSeg) * push.size(), hipMemcpyHostToDevice, ctx->s_main));
    if (!pull.empty()) KR_HIP(hipMemcpyAsync(hp.d_pull, pull.data(), sizeof(HaloPullSeg) * pull.size(), hipMemcpyHostToDevice, ctx->s_main));
    KR_HIP(hipStreamSynchronize(ctx->s_main));
    hp.on = true;
    return halo_peer_selftest(a);
}

// One real exchange over the freshly mapped buffers before anybody relies on them: x[i] = this rank's global row number, pushed, pulled
// and compared with the plan's column list on every rank; the verdict is agreed through an all-gather, so the peer stores are on everywhere
// or nowhere (KRYST_UNSUPPORTED: the RCCL exchange stays).  A mapping that "succeeds" but does not carry the stores -- the one failure the
// set-up itself cannot see -- ends here with a short poll budget instead of in a user's solve.  KRYST_HALO_SELFTEST=0 skips it.
__global__ void halo_iota_kernel(double* x, int64_t n, int64_t base) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) x[i] = (double)(base + i);
}
static int32_t halo_finish(kryst_csr_t a, int budget);
int32_t halo_peer_selftest(kryst_csr_t a) {
    kryst_ctx_t ctx = a->ctx;
    HaloPlan& pl = a->plan;
    if (env_int("KRYST_HALO_SELFTEST", 1) == 0) return KRYST_OK;
    const int P = ctx->nranks;
    int64_t ok = 1;
    double* x = nullptr;
    std::vector<double> got((size_t)pl.total_recv);
    const bool have_cols = (int64_t)pl.recv_cols.size() == pl.total_recv;
    const size_t xbytes = sizeof(double) * (size_t)((a->xlen + KR_TILE - 1) / KR_TILE * KR_TILE + KR_TILE);
    if (hipMalloc(&x, xbytes) != hipSuccess) { (void)hipGetLastError(); ok = 0; x = nullptr; }
    // (a rank that could not allocate still takes part: its neighbours' pulls need its push -- of anything -- and everybody its vote)
    double* src = x ? x : pl.d_halo;                                    // never dereferenced beyond the plan's rows when ok; garbage otherwise
    if (x) {
        hipLaunchKernelGGL(halo_iota_kernel, dim3((unsigned)((a->xlen + 255) / 256)), dim3(256), 0, ctx->s_main, x, a->xlen, pl.row_lo);
        if (hipGetLastError() != hipSuccess) { (void)hipGetLastError(); ok = 0; }
    }
    int32_t rc = KRYST_OK;
    if (x) {
        rc = halo_begin(a, src);
        if (rc == KRYST_OK) rc = halo_finish(a, 1 << 24);     // (a short budget by the solvers' standards -- 2^26 -- yet long enough for peers that time-slice one GPU in the tests)
    } else {
        ++pl.peer.epoch;                                                // keep the exchange count in step with the other ranks
    }
    a->halo_started_for = nullptr;
    if (rc == KRYST_OK && (hipStreamSynchronize(ctx->s_comm) != hipSuccess || hipStreamSynchronize(ctx->s_main) != hipSuccess)) { (void)hipGetLastError(); ok = 0; }
    if (rc != KRYST_OK) ok = 0;
    if (fold_gave_up(ctx)) ok = 0;                                      // (reads and clears the error word of the pull)
    if (ok && pl.total_recv > 0) {
        if (hipMemcpyAsync(got.data(), pl.d_halo, sizeof(double) * got.size(), hipMemcpyDeviceToHost, ctx->s_main) != hipSuccess ||
            hipStreamSynchronize(ctx->s_main) != hipSuccess) { (void)hipGetLastError(); ok = 0; }
        for (int p = 0; p < P && ok; ++p) {
            const int64_t lo = a->row_offsets.size() > (size_t)p + 1 ? a->row_offsets[(size_t)p] : 0, hi = a->row_offsets.size() > (size_t)p + 1 ? a->row_offsets[(size_t)p + 1] : (1ll << 62);
            for (int64_t k = 0; k < pl.recv_counts[p] && ok; ++k) {
                const double v = got[(size_t)(pl.recv_off[p] + k)];
                if (have_cols) ok = v == (double)pl.recv_cols[(size_t)(pl.recv_off[p] + k)];
                else ok = v >= (double)lo && v < (double)hi && (k == 0 || v > got[(size_t)(pl.recv_off[p] + k - 1)]);      // rows of rank p, ascending
            }
        }
    }
    (void)hipFree(x);
    (void)hipMemsetAsync(pl.d_halo, 0, sizeof(double) * (size_t)(pl.total_recv + 2), ctx->s_main);
    // the agreed verdict
    int64_t *d_s = nullptr, *d_r = nullptr;
    std::vector<int64_t> all((size_t)P, 0);
    rc = KRYST_OK;
    if (hipMalloc(&d_s, 8) != hipSuccess || hipMalloc(&d_r, sizeof(int64_t) * P) != hipSuccess) rc = KRYST_ERR_HIP;
    if (rc == KRYST_OK && (hipMemcpyAsync(d_s, &ok, 8, hipMemcpyHostToDevice, ctx->s_main) != hipSuccess || hipStreamSynchronize(ctx->s_main) != hipSuccess)) rc = KRYST_ERR_HIP;
    if (rc == KRYST_OK) rc = comm_all_gather_i64(ctx, d_s, d_r, 1, ctx->s_main);
    if (rc == KRYST_OK && (hipMemcpyAsync(all.data(), d_r, sizeof(int64_t) * P, hipMemcpyDeviceToHost, ctx->s_main) != hipSuccess || hipStreamSynchronize(ctx->s_main) != hipSuccess)) rc = KRYST_ERR_HIP;
    (void)hipFree(d_s); (void)hipFree(d_r);
    if (rc != KRYST_OK) { (void)hipGetLastError(); a->plan.peer.on = false; return rc; }
    for (int p = 0; p < P; ++p)
        if (all[(size_t)p] != 1) {
            a->plan.peer.on = false;                                    // (the buffers stay mapped; a later kryst_csr_halo_mode(a, 1) tests again)
            a->plan.peer.failed = true;
            set_error("halo by peer stores: the test exchange did not arrive intact on rank %d; the RCCL exchange stays in use", p);
            return KRYST_UNSUPPORTED;
        }
    return KRYST_OK;
}

// messages up to this many bytes in total are pushed from the compute stream itself (no second stream, no event): below it the two
// cross-stream dependencies cost more than the stores' time on the wire
static int64_t halo_inline_bytes() { return env_ll("KRYST_HALO_INLINE_BYTES", 256 << 10); }

int32_t halo_begin(kryst_csr_t a, const double* x) {
    kryst_ctx_t ctx = a->ctx;
    HaloPlan& pl = a->plan;
    if (pl.peer.on) {
        HaloPeer& hp = pl.peer;
        ++hp.epoch;
        const bool inl = pl.total_send * 8 <= halo_inline_bytes();
        hipStream_t s = inl ? ctx->s_main : ctx->s_comm;
        if (!inl) {
            KR_HIP(hipEventRecord(ctx->ev_x_ready, ctx->s_main));
            KR_HIP(hipStreamWaitEvent(ctx->s_comm, ctx->ev_x_ready, 0));
        }
        if (hp.npush > 0) {
            const unsigned gx = (unsigned)std::min<int64_t>(std::max<int64_t>(1, (hp.push_max + 1023) / 1024), 64);   // <= 64 workgroups per neighbour: the wire, not the CUs, bounds it
            hipLaunchKernelGGL(halo_push_kernel, dim3(gx, (unsigned)hp.npush), dim3(256), 0, s, x, a->send_contiguous ? nullptr : pl.d_send_idx, hp.d_push, hp.npush,
                               (int)(hp.epoch & 1ull), hp.epoch, hp.d_ticket);
            KR_HIP(hipGetLastError());
        }
        if (!inl) KR_HIP(hipEventRecord(ctx->ev_halo_done, ctx->s_comm));
        a->halo_pushed_inline = inl;
        a->halo_started_for = x;
        return KRYST_OK;
    }
    KR_HIP(hipEventRecord(ctx->ev_x_ready, ctx->s_main));
    KR_HIP(hipStreamWaitEvent(ctx->s_comm, ctx->ev_x_ready, 0));
    const void* sendbase = pl.d_sendbuf;
    if (a->send_contiguous) {
        // k-slab stencils: every send list is a contiguous run of x; ship it in place
        // (send_off then holds the first local row of each run, see kryst_csr_create_dist)
        sendbase = x;
    } else if (pl.total_send > 0) {
        hipLaunchKernelGGL(pack_kernel, dim3((unsigned)((pl.total_send + 255) / 256)), dim3(256), 0, ctx->s_comm,
                           x, pl.d_send_idx, pl.d_sendbuf, pl.total_send);
        KR_HIP(hipGetLastError());
    }
    KR_TRY(comm_exchange(ctx, sendbase, pl.send_counts.data(), pl.send_off.data(), pl.d_halo, pl.recv_counts.data(),
                         pl.recv_off.data(), true, ctx->s_comm));
    KR_HIP(hipEventRecord(ctx->ev_halo_done, ctx->s_comm));
    a->halo_started_for = x;
    return KRYST_OK;
}

// Can the operator take the fused form (spmv_pattern_fuse_kernel)?  A single-rank stencil operator in its CSR-P16 form whose bases are all
// (far, -n, -1, 0, +1, +n, far) with the same far offsets everywhere -- what launch_tiles would hand to spmv_pattern_stage_kernel<.., UFAR>.
bool spmv_can_fuse_direction(kryst_csr_t a) {
    return !a->dist && a->d_pid && a->pat_stage_n > 0 && a->pat_far_uniform && a->npat <= 512 && a->ntab <= 512 && a->nrows == a->xlen &&
           takes_pattern_path(a, false) && env_int("KRYST_SPMV_STAGE", 1) != 0 && env_int("KRYST_SPMV_STAGE_UFAR", 1) != 0 &&
           a->xlen + 2 * KR_TILE < (1ll << 31) &&
           // measured (profiles/r05/cg_fuse_ab.jsonl, cg_fuse_sweep.jsonl, cg_xbatch_ab.jsonl): fewer bytes only pay where the vectors are far beyond
           // the 256 MiB Infinity Cache -- with x in batches 512^3 CG +11 % (PCG +10 %), 480^3 +6 %, 448^3 +-1 %, 416^3 -1 %, 384^3 -6 %, 256^3 0 .. -5 %.
           // KRYST_CG_FUSE_P = 1 / 0 forces the fused / unfused form.
           env_int("KRYST_CG_FUSE_P", a->nrows * 8 > env_ll("KRYST_CG_FUSE_MIN_BYTES", 768ll << 20) ? 1 : 0) != 0;
}
// y = A p_new with p_new = z + beta p_old formed on the way (stored to p_new), the owed x += alpha p_old on the same pass, partial (p_new, y)
// [and (y, y)] -- see spmv_pattern_fuse_kernel.  `done`, alpha, beta, xpend: device scalars of the solve; it: the iteration being enqueued.
static int32_t launch_spmv_fused_impl(kryst_csr_t a, const double* z, const double* p_old, double* p_new, double* xvec, double* y, int nq,
                                      const double* alpha, const double* beta, const long long* xpend, long long it, const int* done, bool force);
int32_t launch_spmv_fused(kryst_csr_t a, const double* z, const double* p_old, double* p_new, double* xvec, double* y, int nq,
                          const double* alpha, const double* beta, const long long* xpend, long long it, const int* done) {
    return launch_spmv_fused_impl(a, z, p_old, p_new, xvec, y, nq, alpha, beta, xpend, it, done, false);
}
static int32_t launch_spmv_fused_impl(kryst_csr_t a, const double* z, const double* p_old, double* p_new, double* xvec, double* y, int nq,
                                      const double* alpha, const double* beta, const long long* xpend, long long it, const int* done, bool force) {
    kryst_ctx_t ctx = a->ctx;
    KR_ARG((force || spmv_can_fuse_direction(a)) && nq >= 1 && nq <= 2, "launch_spmv_fused: operator cannot take the fused form");
    KR_TRY(ensure_partials(ctx, a->ntiles));
    SpmvArgs args;
    memset(&args, 0, sizeof args);
    args.x = p_new; args.y = y; args.ntiles = (int32_t)a->ntiles; args.nrows = (int32_t)a->nrows; args.nloc = (int32_t)a->nrows;
    args.dvec = p_new; args.partials = ctx->d_partials; args.pstride = ctx->partials_cap; args.done = done;
    args.pid = a->d_pid; args.pmeta = a->d_pmeta; args.poff = a->d_poff; args.pval = a->d_pval; args.npat = a->npat; args.ntab = a->ntab;
    const int T = env_int("KRYST_SPMV_FUSE_T", 4) <= 2 ? 2 : 4;     // runs of 4 tiles: the window's halo (2 n + 4 elements) is read for 2 048 rows instead of 1 024
    const int32_t n_ = a->pat_stage_n;
    const size_t xs_bytes = sizeof(double) * (size_t)(T * KR_TILE + 2 * n_ + 4);
    const size_t tab = xs_bytes + (size_t)a->npat * 8 + (size_t)a->ntab * 8;
    args.pat_red_off = (int32_t)((tab + 15) & ~(size_t)15);
    size_t lds_s = (size_t)args.pat_red_off + sizeof(double) * (size_t)T * (size_t)nq * (KR_T / 64);
    // resident workgroups per CU: like the BLAS-1 streams (ew.h: 2-3 workgroups per CU keep DRAM pages open, 8 lose a sixth of the bandwidth) this
    // kernel is a mix of read and write streams; the dynamic LDS request is padded so that only `wgcu` workgroups fit a CU's 160 KiB
    const int wgcu = env_int("KRYST_SPMV_FUSE_WG_PER_CU", 0);
    if (wgcu > 0) lds_s = std::max(lds_s, std::min<size_t>((size_t)(160 << 10) / (size_t)wgcu - 512, (size_t)64 << 10));
    args.xsafe = (a->xlen + KR_TILE - 1) / KR_TILE * KR_TILE + KR_TILE - 2;
    const int64_t nruns = (a->ntiles + T - 1) / T;
    args.group = std::max(1, env_int("KRYST_SPMV_STAGE_GROUP", stage_group_default(a, nruns, T, true)));
    const int64_t per_xcd = ((nruns + 7) / 8 + args.group - 1) / args.group * args.group;
    const dim3 sgrid((unsigned)(per_xcd * 8)), block(KR_T);
    const FuseArgs f{z, p_old, p_new, xvec, alpha, beta, xpend, it};
#ifdef KR_TUNING
    args.abl = env_int("KRYST_FUSE_ABL", 0);          // timing-only ablations (tuning builds; wrong results)
#endif
    if (lds_s > ((size_t)48 << 10)) {       // (more than the default dynamic LDS limit: once per instance)
        static bool raised = false;
        if (!raised) {
#define KR_FUSE_FNS(NQ_, T_, X_) (const void*)spmv_pattern_fuse_kernel<NQ_, T_, 512, X_>, (const void*)spmv_pattern_fuse_kernel<NQ_, T_, 1024, X_>
            for (const void* fn : {KR_FUSE_FNS(1, 2, true), KR_FUSE_FNS(1, 4, true), KR_FUSE_FNS(2, 2, true), KR_FUSE_FNS(2, 4, true),
                                   KR_FUSE_FNS(1, 2, false), KR_FUSE_FNS(1, 4, false), KR_FUSE_FNS(2, 2, false), KR_FUSE_FNS(2, 4, false)})
                (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 64 << 10);
#undef KR_FUSE_FNS
            raised = true;
        }
    }
#define KR_FUSE_N(NQ_, T_, X_) do { if (n_ <= 512) hipLaunchKernelGGL((spmv_pattern_fuse_kernel<NQ_, T_, 512, X_>), sgrid, block, lds_s, ctx->s_main, args, f, n_, a->pat_far_lo, a->pat_far_hi); \
                                    else hipLaunchKernelGGL((spmv_pattern_fuse_kernel<NQ_, T_, 1024, X_>), sgrid, block, lds_s, ctx->s_main, args, f, n_, a->pat_far_lo, a->pat_far_hi); } while (0)
#define KR_FUSE(NQ_) do { if (xvec) { if (T == 2) KR_FUSE_N(NQ_, 2, true); else KR_FUSE_N(NQ_, 4, true); } \
                          else { if (T == 2) KR_FUSE_N(NQ_, 2, false); else KR_FUSE_N(NQ_, 4, false); } } while (0)
    if (nq == 1) KR_FUSE(1); else KR_FUSE(2);
#undef KR_FUSE
#undef KR_FUSE_N
    KR_HIP(hipGetLastError());
    phase_mark(ctx, KR_PH_SPMV);
    return KRYST_OK;
}

// the receiving half of an exchange halo_begin started: afterwards the compute stream may read the plan's d_halo (budget <= 0: the default poll budget)
static int32_t halo_finish(kryst_csr_t a, int budget) {
    kryst_ctx_t ctx = a->ctx;
    if (a->plan.peer.on) {
        HaloPeer& hp = a->plan.peer;
        // (my own push has read x: only needed before x is overwritten, and long since true -- the one local dependency that is left)
        if (!a->halo_pushed_inline) KR_HIP(hipStreamWaitEvent(ctx->s_main, ctx->ev_halo_done, 0));
        if (hp.npull > 0) {
            static const int dflt = [] { const char* e = getenv("KRYST_IPC_POLL_BUDGET"); return e ? std::max(1, atoi(e)) : (1 << 26); }();
            const unsigned gx = (unsigned)std::min<int64_t>(std::max<int64_t>(1, (hp.pull_max + 1023) / 1024), 256);
            hipLaunchKernelGGL(halo_pull_kernel, dim3(gx, (unsigned)hp.npull), dim3(256), 0, ctx->s_main, hp.landing + (size_t)(hp.epoch & 1ull) * hp.stride,
                               a->plan.d_halo, hp.d_pull, hp.epoch, budget > 0 ? std::min(budget, dflt) : dflt, fold_err(ctx) + 1);
            KR_HIP(hipGetLastError());
        }
    } else {
        KR_HIP(hipStreamWaitEvent(ctx->s_main, ctx->ev_halo_done, 0));
    }
    return KRYST_OK;
}

int32_t launch_spmv(kryst_csr_t a, const double* x, double* y, int nq, const double* dvec, const int* done) {
    kryst_ctx_t ctx = a->ctx;
    if (nq > 0) KR_TRY(ensure_partials(ctx, a->ntiles));
    if (!a->dist || !use_collectives(ctx)) {
        const int32_t rc = launch_tiles<false>(a, x, y, nq, dvec, done, nullptr, a->ntiles);
        phase_mark(ctx, KR_PH_SPMV);
        return rc;
    }
    // halo exchange on s_comm (unless the solver has started it already), overlapped with the interior tiles
    if (a->halo_started_for != x) KR_TRY(halo_begin(a, x));
    a->halo_started_for = nullptr;
    KR_TRY(launch_tiles<false>(a, x, y, nq, dvec, done, a->d_tiles_interior, a->n_interior));   // interior tiles have no halo columns
    phase_mark(ctx, KR_PH_SPMV);
    KR_TRY(halo_finish(a, 0));
    phase_mark(ctx, KR_PH_HALO_WAIT);
    KR_TRY(launch_tiles<true>(a, x, y, nq, dvec, done, a->d_tiles_boundary, a->n_boundary));
    phase_mark(ctx, KR_PH_SPMV_BOUNDARY);
    return KRYST_OK;
}

}  // namespace kr

using namespace kr;

extern "C" {

// How the operator's halo exchange travels: mode 0 = grouped ncclSend / ncclRecv on the second stream (default), 1 = direct peer stores
// into hipIpc-mapped landing buffers (dist.h: HaloPeer).  COLLECTIVE: every rank calls it with the same mode.  KRYST_UNSUPPORTED (mode 1)
// when some rank cannot export / map a landing buffer or a neighbour relation is one-way: every rank then stays on RCCL.  *active: the
// mode in use.  The results are bit-identical either way (the same values land in the same halo slots).
int32_t kryst_csr_halo_mode(kryst_csr_t a, int32_t mode, int32_t* active);
}
namespace kr {
// A freshly created row-partitioned operator takes the peer-store exchange when every rank can map its neighbours' landing buffers and the
// test exchange arrives intact everywhere (agreed outcome); otherwise -- not an error -- the grouped ncclSend / ncclRecv exchange stays.
// KRYST_HALO_MODE=rccl: do not try.  COLLECTIVE like the creation itself.
int32_t halo_default_mode(kryst_csr_t a) {
    if (!a->dist || !a->ctx->comm || a->ctx->nranks < 2) return KRYST_OK;
    const char* e = getenv("KRYST_HALO_MODE");
    if (e && strcmp(e, "rccl") == 0) return KRYST_OK;
    const int32_t rc = kryst_csr_halo_mode(a, 1, nullptr);
    return rc == KRYST_UNSUPPORTED ? KRYST_OK : rc;
}
}
extern "C" {
int32_t kryst_csr_halo_mode(kryst_csr_t a, int32_t mode, int32_t* active) {
    KR_ARG(a && (mode == 0 || mode == 1 || mode == -1), "csr_halo_mode");
    if (mode == -1) { if (active) *active = a->plan.peer.on ? 1 : 0; return KRYST_OK; }       // query only
    KR_ARG(a->ctx->active_ws == nullptr, "csr_halo_mode: a solve or stepping session is open on this context");
    int32_t rc = KRYST_OK;
    if (a->dist && a->ctx->comm) {
        KR_HIP(hipSetDevice(a->ctx->device));
        KR_HIP(hipStreamSynchronize(a->ctx->s_comm));
        KR_HIP(hipStreamSynchronize(a->ctx->s_main));
        a->halo_started_for = nullptr;
        // a pull that gave up earlier must not hand its exhausted patience to the mode chosen now (the word is sticky until the host reads it)
        KR_HIP(hipMemsetAsync(fold_err(a->ctx) + 1, 0, sizeof(unsigned int), a->ctx->s_main));
        KR_HIP(hipStreamSynchronize(a->ctx->s_main));
        if (mode == 1) rc = halo_peer_setup(a);
        else a->plan.peer.on = false;            // (the landing buffers stay mapped: switching back costs nothing)
    }
    if (active) *active = a->plan.peer.on ? 1 : 0;
    return rc;
}

int32_t kryst_spmv(kryst_csr_t a, kryst_vec_t x, kryst_vec_t y) {
    KR_ARG(a && x && y, "spmv");
    KR_ARG(x->ctx == a->ctx && y->ctx == a->ctx, "spmv: context mismatch");
    KR_ARG(x->n == a->xlen, "spmv: x.len() != ncols");      // sparse.rs:57 assert_eq!
    KR_ARG(y->n == a->nrows, "spmv: y.len() != nrows");     // sparse.rs:58 assert_eq!
    KR_ARG(x->d != y->d, "spmv: x and y alias");
    KR_HIP(hipSetDevice(a->ctx->device));
    KR_TRY(launch_spmv(a, x->d, y->d, 0, nullptr, nullptr));
    // outside a solver nobody else reads the pull's error word: a neighbour's stamp that never came (the halo is NaNs then) is this call's error
    if (a->dist && a->plan.peer.on && use_collectives(a->ctx) && fold_gave_up(a->ctx)) return KRYST_ERR_RCCL;
    return KRYST_OK;
}

// which encoding kryst_spmv streams for this operator under the current KRYST_SPMV_COMPRESS setting:
// 0 plain CSR (12 B/nnz), 1 CSR-D8 (9 B/nnz), 2 CSR-D16 (2 B/nnz), 3 CSR-P16 (2 B/row), 4 CSR-DIA (8 B per diagonal and row)
int32_t kryst_csr_encoding(kryst_csr_t a, int32_t* encoding, int32_t* patterns, int32_t* table_entries) {
    KR_ARG(a && encoding, "csr_encoding");
    const int lvl = env_int("KRYST_SPMV_COMPRESS", 3);
    int e = 0;
    const bool p16_fits = a->xlen + a->plan.total_recv < (1ll << 28) && a->nrows < (1ll << 28);     // (32-bit byte offsets: launch_tiles)
    if (a->d_pid && lvl >= 3 && p16_fits) e = 3;
    else if (a->d_code16 && lvl >= 2) e = 2;
    else if (a->d_dia && lvl >= 1 && env_int("KRYST_SPMV_DIA", 1) != 0) e = 4;
    else if (a->d_code && lvl >= 1) e = 1;
    *encoding = e;
    if (patterns) *patterns = e == 4 ? a->dia_nd : (a->d_pid ? a->npat : 0);      // CSR-DIA: the number of diagonals
    if (table_entries) *table_entries = a->d_pid ? a->ntab : 0;
    return KRYST_OK;
}

// info[0]: rows per plane of the slab order (0: the operator has none), info[1] / info[2]: tile slots in the orders for runs of 1 / 8
// slots per XCD, info[3]: 1 when kryst_spmv would use the order under the current settings
int32_t kryst_csr_tile_order(kryst_csr_t a, int64_t* info) {
    KR_ARG(a && info, "csr_tile_order");
    info[0] = a->d_tile_order ? a->order_plane : 0; info[1] = a->order_slots1; info[2] = a->order_slots8;
    info[3] = uses_tile_order(a) ? 1 : 0;
    return KRYST_OK;
}

// info[0]: line length n of the staged-window form of the CSR-P16 kernel (0: the operator's bases are not (far, -n, -1, 0, +1, +n, far)),
// info[1]: 1 when every base has the same far offsets (one round trip per run), info[2]: first tile of a rank's contiguous interior
// range (-1: none), info[3]: 1 when kryst_spmv would take the staged-window kernel under the current settings
int32_t kryst_csr_pattern_info(kryst_csr_t a, int64_t* info) {
    KR_ARG(a && info, "csr_pattern_info");
    info[0] = a->d_pid ? a->pat_stage_n : 0; info[1] = a->pat_far_uniform ? 1 : 0; info[2] = a->interior_first;
    info[3] = (a->d_pid && a->pat_stage_n > 0 && takes_pattern_path(a, false) && a->npat <= 512 && a->ntab <= 512 && env_int("KRYST_SPMV_STAGE", 1) != 0 &&
               a->xlen + 2 * KR_TILE < (1ll << 31) && (!a->dist || !use_collectives(a->ctx) || a->interior_first >= 0)) ? 1 : 0;
    return KRYST_OK;
}

int32_t kryst_bench_spmv(kryst_csr_t a, kryst_vec_t x, kryst_vec_t y, int32_t fused_dots, int32_t reps, double* avg_ms) {
    KR_ARG(a && x && y && avg_ms && reps >= 1 && fused_dots >= 0 && fused_dots <= 2, "bench_spmv");
    KR_ARG(x->n == a->xlen && y->n == a->nrows, "bench_spmv: size mismatch");
    KR_ARG(fused_dots == 0 || a->nrows == a->xlen, "bench_spmv: fused dots need a square operator");
    kryst_ctx_t ctx = a->ctx;
    KR_HIP(hipSetDevice(ctx->device));
    // rotate over three copies of x: inside a solver the input vector has just been rewritten and does not sit in the
    // 256 MiB Infinity Cache, which a loop over ONE 128 MiB vector would (256^3: 63 us cache-warm against 86 us in CG)
    const size_t bytes = sizeof(double) * (size_t)((x->n + KR_TILE - 1) / KR_TILE * KR_TILE + KR_TILE);
    double* xs[3] = {x->d, nullptr, nullptr};
    for (int k = 1; k < 3; ++k) {
        if (hipMalloc(&xs[k], bytes) != hipSuccess) { (void)hipGetLastError(); xs[k] = x->d; continue; }
        KR_HIP(hipMemcpyAsync(xs[k], x->d, bytes, hipMemcpyDeviceToDevice, ctx->s_main));
    }
    int32_t rc = launch_spmv(a, xs[0], y->d, fused_dots, xs[0], nullptr);        // warm-up launch
    (void)hipEventRecord(ctx->tm0, ctx->s_main);
    for (int r = 0; r < reps && rc == KRYST_OK; ++r) rc = launch_spmv(a, xs[(r + 1) % 3], y->d, fused_dots, xs[(r + 1) % 3], nullptr);
    (void)hipEventRecord(ctx->tm1, ctx->s_main);
    (void)hipEventSynchronize(ctx->tm1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, ctx->tm0, ctx->tm1);
    *avg_ms = (double)ms / reps;
    (void)hipStreamSynchronize(ctx->s_main);
    if (use_collectives(ctx)) (void)hipStreamSynchronize(ctx->s_comm);
    for (int k = 1; k < 3; ++k) if (xs[k] != x->d) (void)hipFree(xs[k]);
    return rc;
}

// Measurement hook: average milliseconds per launch of the fused direction + SpMV kernel of CG / PCG (spmv_pattern_fuse_kernel) on work vectors of its
// own: z = x, p_old = a copy of x, alpha = 1e-3, beta = 0.5, the x update owed at every launch; p_old / p_new alternate like in the solver.
// KRYST_UNSUPPORTED when the operator cannot take the fused form.
int32_t kryst_bench_spmv_fused(kryst_csr_t a, kryst_vec_t x, kryst_vec_t y, int32_t reps, double* avg_ms) {
    KR_ARG(a && x && y && avg_ms && reps >= 1, "bench_spmv_fused");
    KR_ARG(x->n == a->xlen && y->n == a->nrows && x->ctx == a->ctx && y->ctx == a->ctx, "bench_spmv_fused: size / context mismatch");
    kryst_ctx_t ctx = a->ctx;
    KR_HIP(hipSetDevice(ctx->device));
    if (!(!a->dist && a->d_pid && a->pat_stage_n > 0 && a->pat_far_uniform && a->npat <= 512 && a->ntab <= 512 && a->nrows == a->xlen && takes_pattern_path(a, false))) {
        set_error("bench_spmv_fused: the operator has no staged CSR-P16 form with uniform far offsets");
        return KRYST_UNSUPPORTED;
    }
    const size_t bytes = sizeof(double) * (size_t)((x->n + KR_TILE - 1) / KR_TILE * KR_TILE + KR_TILE);
    double *p[2] = {nullptr, nullptr}, *xv = nullptr, *sc = nullptr;
    int32_t rc = KRYST_OK;
    if (hipMalloc(&p[0], bytes) != hipSuccess || hipMalloc(&p[1], bytes) != hipSuccess || hipMalloc(&xv, bytes) != hipSuccess || hipMalloc(&sc, 64) != hipSuccess) rc = KRYST_ERR_HIP;
    const double host_sc[4] = {1e-3, 0.5, 0.0, 0.0};                 // alpha, beta, xpend (as a long long 0), pad
    if (rc == KRYST_OK && (hipMemcpyAsync(p[0], x->d, bytes, hipMemcpyDeviceToDevice, ctx->s_main) != hipSuccess || hipMemsetAsync(p[1], 0, bytes, ctx->s_main) != hipSuccess ||
                           hipMemsetAsync(xv, 0, bytes, ctx->s_main) != hipSuccess || hipMemcpyAsync(sc, host_sc, sizeof host_sc, hipMemcpyHostToDevice, ctx->s_main) != hipSuccess ||
                           hipStreamSynchronize(ctx->s_main) != hipSuccess)) rc = KRYST_ERR_HIP;
    const long long* xpend = reinterpret_cast<const long long*>(sc + 2);
    auto once = [&](int r) { return launch_spmv_fused_impl(a, x->d, p[r & 1], p[(r + 1) & 1], xv, y->d, 1, sc, sc + 1, xpend, 1ll, nullptr, true); };
    if (rc == KRYST_OK) rc = once(0);
    if (rc == KRYST_OK) {
        (void)hipEventRecord(ctx->tm0, ctx->s_main);
        for (int r = 1; r <= reps && rc == KRYST_OK; ++r) rc = once(r);
        (void)hipEventRecord(ctx->tm1, ctx->s_main);
        (void)hipEventSynchronize(ctx->tm1);
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, ctx->tm0, ctx->tm1);
        *avg_ms = (double)ms / reps;
    }
    (void)hipStreamSynchronize(ctx->s_main);
    (void)hipFree(p[0]); (void)hipFree(p[1]); (void)hipFree(xv); (void)hipFree(sc);
    if (rc == KRYST_ERR_HIP) set_error("bench_spmv_fused: allocation or copy failed");
    return rc;
}

int32_t kryst_spmv_host(kryst_csr_t a, const double* x, int64_t nx, double* y, int64_t ny) {
    KR_ARG(a && x && y, "spmv_host");
    kryst_vec_t vx = nullptr, vy = nullptr;
    KR_TRY(kryst_vec_create(a->ctx, nx, &vx));
    int32_t rc = kryst_vec_create(a->ctx, ny, &vy);
    if (rc == KRYST_OK) rc = kryst_vec_upload(vx, x, nx);
    if (rc == KRYST_OK) rc = kryst_spmv(a, vx, vy);
    if (rc == KRYST_OK) rc = kryst_vec_download(vy, y, ny);
    kryst_vec_destroy(vx); kryst_vec_destroy(vy);
    return rc;
}

}  // extern "C"
