// Tile-structured elementwise kernels with optional fused inner-product partials.
//
// Every vector kernel walks the vector in tiles of KR_TILE = KR_T*KR_V elements; thread t of a tile owns the
// KR_V = 2 consecutive elements q*512 + 2t, +1 (one 16-byte access per array: the coalescing sweet spot on
// gfx950).  A fused reduction folds the thread's two terms in index order, then the 64-lane xor butterfly,
// then the 4 waves serially, and stores ONE partial per tile: the association order depends only on the
// element index, never on the grid, so fusing a dot into any kernel gives the same bits as the standalone dot.
//
// Vectors are allocated padded to a multiple of KR_TILE (zero filled), so a tile is always fully addressable;
// reductions mask i < n.
#pragma once
#include "common.h"
#include <algorithm>
#include <cstdio>
#include <cstdlib>

namespace kr {

// KR_NT: bit 0 = nontemporal vector loads, bit 1 = nontemporal vector stores in the BLAS-1 streams (measured in
// tools/nt_bench.py)
#ifndef KR_NT
#define KR_NT 3
#endif
typedef double kr_v2d __attribute__((ext_vector_type(2)));
struct d2 { double a, b; };
__device__ __forceinline__ d2 ld2(const double* p, int64_t i) {
#if KR_NT & 1
    const kr_v2d v = __builtin_nontemporal_load(reinterpret_cast<const kr_v2d*>(p + i));
#else
    const kr_v2d v = *reinterpret_cast<const kr_v2d*>(p + i);
#endif
    return {v.x, v.y};
}
__device__ __forceinline__ void st2(double* p, int64_t i, double a, double b) {
    kr_v2d v; v.x = a; v.y = b;
#if KR_NT & 2
    __builtin_nontemporal_store(v, reinterpret_cast<kr_v2d*>(p + i));
#else
    *reinterpret_cast<kr_v2d*>(p + i) = v;
#endif
}
// temporal (cacheable) accesses for a vector that the NEXT launch reads again
__device__ __forceinline__ d2 ld2_keep(const double* p, int64_t i) {
    const kr_v2d v = *reinterpret_cast<const kr_v2d*>(p + i);
    return {v.x, v.y};
}
__device__ __forceinline__ void st2_keep(double* p, int64_t i, double a, double b) {
    kr_v2d v; v.x = a; v.y = b;
    *reinterpret_cast<kr_v2d*>(p + i) = v;
}
// compile-time choice per kernel instance: keep_in_cache(n) (solver_common.h) picks the instance at launch
template <bool KEEP> __device__ __forceinline__ d2 ld2_sel(const double* p, int64_t i) { if constexpr (KEEP) return ld2_keep(p, i); else return ld2(p, i); }
template <bool KEEP> __device__ __forceinline__ void st2_sel(double* p, int64_t i, double a, double b) { if constexpr (KEEP) st2_keep(p, i, a, b); else st2(p, i, a, b); }
// coefficient that lives either in a kernel argument or in device memory (written by a scalar kernel)
struct Coef {
    const double* ptr; double val;
    __device__ __forceinline__ double get() const { return ptr ? *ptr : val; }
};
inline Coef coef_dev(const double* p) { return Coef{p, 0.0}; }
inline Coef coef_val(double v) { return Coef{nullptr, v}; }

// Op requirements:  static constexpr int NQ;  __device__ void pair(int64_t i, bool in0, bool in1, double (&acc)[max(NQ,1)]) const;
// Gate: decides, uniformly for the grid, whether the launch is a no-op (the solver has ended, the restart cycle has been left, ...)
struct GateDone {                    // `done` flag of the solve (nullptr: always run)
    const int* done;
    __device__ __forceinline__ bool skip() const { return done && *done; }
};

template <class Op, class Gate>
__global__ __launch_bounds__(KR_T) void ew_kernel(Op op, Gate gate, int64_t n, int64_t tile_lo, int64_t ntiles, double* partials, int64_t pstride) {
    if (gate.skip()) return;
    constexpr int NQ = Op::NQ;
    __shared__ double lds[(NQ > 0 ? NQ : 1) * (KR_T / 64)];
    for (int64_t q = tile_lo + blockIdx.x; q < ntiles; q += gridDim.x) {          // tiles [tile_lo, ntiles): the whole vector, or a range of it
        const int64_t i = q * KR_TILE + (int64_t)threadIdx.x * KR_V;
        double acc[NQ > 0 ? NQ : 1];
#pragma unroll
        for (int k = 0; k < (NQ > 0 ? NQ : 1); ++k) acc[k] = 0.0;
        op.pair(i, i < n, i + 1 < n, acc);
        if constexpr (NQ > 0) {
            block_reduce_any<NQ, KR_T / 64>(acc, lds);
            if (threadIdx.x == 0) {
#pragma unroll
                for (int k = 0; k < NQ; ++k) partials[k * pstride + q] = acc[k];
            }
        }
    }
}

// workgroups per CU of the capped grid: 2 for kernels that also write; read-only reductions declare BPC = 4
template <class Op, class = void> struct ew_bpc { static constexpr int value = 2; };
template <class Op> struct ew_bpc<Op, std::void_t<decltype(Op::BPC)>> { static constexpr int value = Op::BPC; };
// tuning: an op that declares `static constexpr const char* TAG = "X"` can have its workgroups per CU set per launch through
// KRYST_BPC_X (tools/solver_ab.py times whole solver iterations with the settings in turn, in one process)
template <class Op, class = void> struct ew_tag { static const char* get() { return nullptr; } };
template <class Op> struct ew_tag<Op, std::void_t<decltype(Op::TAG)>> { static const char* get() { return Op::TAG; } };

// phase an op's launches are charged to while kryst_phase_timing is on: KR_PH_BLAS1 unless the op declares `static constexpr int PHASE`
template <class Op, class = void> struct ew_phase { static constexpr int value = KR_PH_BLAS1; };
template <class Op> struct ew_phase<Op, std::void_t<decltype(Op::PHASE)>> { static constexpr int value = Op::PHASE; };

// bpc <= 0: the kernel shape's default (KRYST_EW_BLOCKS_PER_CU overrides it)
template <class Op, class Gate>
inline int32_t launch_ew_gated(kryst_ctx_t ctx, const Op& op, int64_t n, const Gate& gate, int bpc = 0, int64_t tile_lo = 0, int64_t tile_hi = -1) {
    const int64_t all_tiles = ntiles_of(n);
    if (tile_hi < 0 || tile_hi > all_tiles) tile_hi = all_tiles;             // [tile_lo, tile_hi): a launch over part of the vector (the partials keep their tile numbers)
    const int64_t ntiles = tile_hi - tile_lo;
    if (ntiles <= 0) return KRYST_OK;
    if (Op::NQ > 0) KR_TRY(ensure_partials(ctx, all_tiles));
    // memory-bound streaming: cap the grid and stride the rest.  Measured on MI355X (tools/stream_bench.py, vectors of
    // 1 GiB, interleaved rounds): 2-3 workgroups per CU sustain 5.6-5.8 TB/s on mixed read/write streams, 8 per CU only
    // 4.7-4.9 TB/s (a narrower moving window keeps DRAM pages open); CG at 512^3: +4 %.  Pure read streams with a
    // reduction per tile (dots) are the exception: they need 4 per CU to overlap loads with the butterfly (rocprofv3,
    // 9 streams of 128 MiB: 420 us at 2 per CU, 250 us at 4).
    if (bpc <= 0) bpc = env_int("KRYST_EW_BLOCKS_PER_CU", ew_bpc<Op>::value);      // tuning knobs: read per launch outside a solve, once per solve / session step inside
    if (const char* tag = ew_tag<Op>::get()) {
        static const std::string name = std::string("KRYST_BPC_") + tag;          // (one name, and one slot, per op type)
        const int t = env_int(name.c_str(), 0);
        if (t > 0) bpc = t;
    }
    const int64_t grid = std::min<int64_t>(ntiles, (int64_t)ctx->num_cu * bpc);
    hipLaunchKernelGGL((ew_kernel<Op, Gate>), dim3((unsigned)grid), dim3(KR_T), 0, ctx->s_main, op, gate, n, tile_lo, tile_hi,
                       ctx->d_partials, ctx->partials_cap);
    KR_HIP(hipGetLastError());
    phase_mark(ctx, ew_phase<Op>::value);
    return KRYST_OK;
}

template <class Op>
inline int32_t launch_ew(kryst_ctx_t ctx, const Op& op, int64_t n, const int* done = nullptr) {
    return launch_ew_gated(ctx, op, n, GateDone{done});
}

}  // namespace kr
