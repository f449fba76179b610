// Internal definitions shared by the libkryst_hip.so translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <type_traits>
#include <vector>
#include "../../include/kryst_hip.h"

// ---- the fixed inner-product tree (kryst_reduce_spec) ----
#define KR_T 256            // threads per tile
#define KR_V 2              // elements per thread per tile
#define KR_TILE (KR_T * KR_V)
#define KR_F 1024           // threads of the final fold
#define KR_MAXQ 8           // at most 8 fused reductions per kernel (FGMRES batches its Gram-Schmidt dots by 8)

#include <atomic>

namespace kr {

// ---- tuning knobs from the environment.  Outside a solve every read is a getenv (a test or tool sets a knob and calls an entry
// point: it takes effect at once).  INSIDE a solver's enqueue loop (EnvFreeze: a one-shot solve, or one kryst_session_step call) a
// knob is read once and then served from its call site's slot -- an iteration used to cost ~30 scans of environ on the host's enqueue
// path, and getenv races with a setenv from another host thread.  The epoch is bumped whenever a freeze begins, so a
// tool that changes a knob between two session steps (tools/solver_ab.py) still sees it.
struct EnvSlot { std::atomic<uint32_t> epoch{0}; std::atomic<int> has{0}; std::atomic<long long> val{0}; };
extern std::atomic<uint32_t> g_env_epoch;        // ctx.cpp
extern thread_local int g_env_frozen;
inline long long env_ll_cached(EnvSlot& s, const char* name, long long dflt) {
    if (g_env_frozen > 0) {
        const uint32_t e = g_env_epoch.load(std::memory_order_relaxed);
        if (s.epoch.load(std::memory_order_acquire) == e) return s.has.load(std::memory_order_relaxed) ? s.val.load(std::memory_order_relaxed) : dflt;
        const char* v = getenv(name);
        s.has.store(v ? 1 : 0, std::memory_order_relaxed); s.val.store(v ? atoll(v) : 0, std::memory_order_relaxed);
        s.epoch.store(e, std::memory_order_release);
        return v ? atoll(v) : dflt;
    }
    const char* v = getenv(name);
    return v ? atoll(v) : dflt;
}
struct EnvFreeze {                               // RAII around a solver's enqueue loop
    EnvFreeze() { if (g_env_frozen++ == 0) g_env_epoch.fetch_add(1, std::memory_order_relaxed); }
    ~EnvFreeze() { --g_env_frozen; }
    EnvFreeze(const EnvFreeze&) = delete; EnvFreeze& operator=(const EnvFreeze&) = delete;
};
// one slot per call site (the name must be a literal or otherwise fixed for the site)
#define env_int(name, dflt) ([&]() -> int { static kr::EnvSlot slot__; return (int)kr::env_ll_cached(slot__, (name), (long long)(dflt)); }())
#define env_ll(name, dflt) ([&]() -> long long { static kr::EnvSlot slot__; return kr::env_ll_cached(slot__, (name), (long long)(dflt)); }())

void set_error(const char* fmt, ...);
void set_error_row(int64_t row);      // the row of a KRYST_ZERO_PIVOT (KError::ZeroPivot(row), src/error.rs:15-16)

#define KR_HIP(call)                                                                          \
    do {                                                                                      \
        hipError_t e__ = (call);                                                              \
        if (e__ != hipSuccess) {                                                              \
            kr::set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e__), __FILE__, __LINE__); \
            return KRYST_ERR_HIP;                                                             \
        }                                                                                     \
    } while (0)

#define KR_TRY(call)                                   \
    do {                                               \
        int32_t rc__ = (call);                         \
        if (rc__ != KRYST_OK) return rc__;             \
    } while (0)

#define KR_ARG(cond, msg)                                          \
    do {                                                           \
        if (!(cond)) { kr::set_error("bad argument: %s", msg); return KRYST_ERR_ARG; } \
    } while (0)

// Progress record in mapped host memory: the device's scalar kernels publish here, the host polls it
// between batches without touching the stream.
struct HostProgress {
    volatile int64_t iter;
    volatile double  res;
    volatile int32_t done;
    volatile int32_t status;
    volatile int64_t hist_len;      // residual-history entries written so far (published only when a monitor is attached)
};

struct Comm;   // dist.cpp

// Optional per-phase timing of the compute stream (kryst_phase_timing_begin / _end; bench.py's `phase_ms`): while it is on, the
// launchers record a hipEvent after each phase of an iteration and the time between two consecutive marks is charged to the
// later mark's phase.  Off (the default) a mark is one pointer test.
enum { KR_PH_SPMV = 0,          // SpMV tiles that need no halo (single rank: the whole SpMV)
       KR_PH_HALO_WAIT,         // compute stream waiting for the halo exchange after the interior tiles
       KR_PH_SPMV_BOUNDARY,     // tiles with halo columns
       KR_PH_REDUCE,            // tile-partial fold, RCCL all-gather, rank-ordered fold + the solver's scalar step
       KR_PH_BLAS1,             // fused vector updates (incl. their tile partials) that are neither of the two below
       KR_PH_PC,                // preconditioner apply
       KR_PH_BLAS1_RESIDUAL,    // CG / PCG: r -= alpha Ap with the fused (r,r) [, z = D^-1 r, (r,z)]   (CgResidualOp / PcgResidualOp / the eager forms)
       KR_PH_BLAS1_DIRECTION,   // CG / PCG: x += alpha p, p = z + beta p                                (CgDirectionOp / AypxDevOp)
       KR_PH_BLAS1_XBATCH,      // CG / PCG with x updated in batches: x += alpha_i p_i for the last m iterations in one pass    (XBatchOp)
       KR_PH_COUNT };
struct PhaseTimer {
    std::vector<std::pair<int, hipEvent_t>> marks;      // (phase, event recorded after it); phase -1: the start mark
    std::vector<hipEvent_t> pool;                       // events to reuse
};

}  // namespace kr

struct kryst_ctx_s {
    int device = 0;
    int rank = 0, nranks = 1;
    hipStream_t s_main = nullptr;   // compute
    hipStream_t s_comm = nullptr;   // halo exchange
    hipEvent_t ev_x_ready = nullptr, ev_halo_done = nullptr, tm0 = nullptr, tm1 = nullptr;
    hipEvent_t ev_ring[4] = {nullptr, nullptr, nullptr, nullptr};
    kr::Comm* comm = nullptr;
    // reduction scratch: KR_MAXQ arrays of tile partials, sized on demand
    double* d_partials = nullptr; int64_t partials_cap = 0;     // doubles per array
    double* d_chunks = nullptr; int64_t chunks_cap = 0;         // stage-1 results of the two-level fold
    unsigned int* d_ticket = nullptr;   // [0]: ticket of the two-level fold, [8]: error word of its polling form (fold_err)
    bool fold_poll_off = false;         // a polling hand-off timed out on this context: ticket form from now on
    double* d_scal = nullptr;        // small scalar arena (device), 4096 doubles
    double* d_gather = nullptr;      // nranks * KR_MAXQ doubles (all-gather target)
    kr::HostProgress* h_prog = nullptr; kr::HostProgress* d_prog = nullptr;   // mapped
    double* h_pinned = nullptr;      // 4096 doubles pinned staging
    // work-vector arena: ONE allocation that the vectors of a solve are carved from and that is kept between solves
    // (separately hipMalloc'ed 128 MiB vectors land wherever the allocator has room, and multi-stream kernels then run
    // up to 30 % slower and vary from solve to solve; a single block does not -- tools/stride_bench.py)
    char* arena = nullptr; size_t arena_bytes = 0, arena_used = 0; const void* arena_owner = nullptr;
    // The scalar state of a solve (DevState, reduction results, progress record) lives in per-context scratch, so ONE solve
    // or stepping session may be open per context at a time: a second one is refused with KRYST_ERR_BUSY (kryst_hip.h).
    const void* active_ws = nullptr;
    // Scalar all-reduce without a collective launch (dist.cpp: ipc_reduce_*): every rank owns a fine-grained mailbox that the
    // others map through hipIpc; the kernel that finishes a rank's local fold stores its partials into every peer's mailbox
    // (self-validating cells: each 8-byte word carries half a value and the epoch's tag), polls its own mailbox and folds in rank order --
    // the same bits as the RCCL all-gather + ordered fold, one launch instead of two launches and a collective.
    double* ipc_mine = nullptr;                  // 2 parities x nranks cells of 16 words: [parity][writer rank]{(low half | tag), (high half | tag)} x 8 quantities (solver_common.h)
    double** d_ipc_peers = nullptr;              // device array: peer p's mailbox as mapped into this process (own entry: ipc_mine)
    std::vector<void*> ipc_opened;               // mappings to close
    unsigned long long* d_ipc_epoch = nullptr;   // reductions completed so far (device)
    bool ipc_on = false;                         // reduce_then uses the mailbox path
    bool ipc_failed = false;                     // the test reduction over the mailboxes failed (agreed across ranks): never switched on again
    kr::PhaseTimer* phase = nullptr;     // non-null while kryst_phase_timing is on
    std::vector<void*> deferred_free;    // device allocations that outlive their owner and go with the context (spmv.hip: halo_peer_destroy)
    int num_cu = 256;
};

struct kryst_vec_s {
    kryst_ctx_t ctx = nullptr;
    int64_t n = 0;
    double* d = nullptr;
};

namespace kr {

int32_t ensure_partials(kryst_ctx_t ctx, int64_t ntiles);

// Device blocks of a destroyed ILU-family preconditioner (factor streams, blocked layouts, work vectors, the set-up's scratch) are kept in a
// size-keyed pool per DEVICE and handed to the next set-up that asks for the same size -- Ilup::setup is called per matrix, repeatedly
// (ilup.rs:77-134), and at 512^3 giving 22 GB back to the driver and taking it again cost ten times the factorisation.
// pool_malloc: an exact-size block from the pool, else hipMalloc (the pool is emptied and the call repeated once when the driver has no room).
// pool_free: waits for the device like hipFree does, then keeps blocks of >= 1 MiB while the pool stays below KRYST_DEV_POOL_MB (default
// 65536; 0: no pool), frees otherwise.  pool_trim gives everything back (kryst_ctx_trim).  Thread-safe.
hipError_t pool_malloc_bytes(void** p, size_t bytes);
hipError_t pool_free(void* p);
size_t pool_trim(int device);                         // -> bytes returned to the driver
template <class T> inline hipError_t pool_malloc(T** p, size_t bytes) { return pool_malloc_bytes(reinterpret_cast<void**>(p), bytes); }
void phase_mark_slow(kryst_ctx_t ctx, int phase);                       // ctx.cpp
inline void phase_mark(kryst_ctx_t ctx, int phase) { if (ctx->phase) phase_mark_slow(ctx, phase); }
inline int64_t ntiles_of(int64_t n) { return (n + KR_TILE - 1) / KR_TILE; }
inline int64_t nchunks_of(int64_t ntiles) { return ntiles > KR_F ? (ntiles + KR_F - 1) / KR_F : 1; }

// ---- device-side reduction primitives (the association order is part of the ABI contract) ----
#ifdef __HIPCC__
__device__ __forceinline__ double wave_butterfly(double v) {
    v = v + __shfl_xor(v, 32, 64);
    v = v + __shfl_xor(v, 16, 64);
    v = v + __shfl_xor(v, 8, 64);
    v = v + __shfl_xor(v, 4, 64);
    v = v + __shfl_xor(v, 2, 64);
    v = v + __shfl_xor(v, 1, 64);
    return v;
}

// NW waves per block; lds must hold NQ*NW doubles.  Result valid in every thread.
template <int NQ, int NW>
__device__ __forceinline__ void block_reduce(double (&v)[NQ], double* lds) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        v[q] = wave_butterfly(v[q]);
        if (lane == 0) lds[q * NW + wave] = v[q];
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        double s = lds[q * NW];
#pragma unroll
        for (int w = 1; w < NW; ++w) s = s + lds[q * NW + w];
        v[q] = s;
    }
    __syncthreads();
}

// Same values as block_reduce for NQ = 2, 4 or 8 quantities with a fifth of the cross-lane traffic: the first log2(NQ)
// butterfly steps exchange HALF of the quantities each (the lane whose `off` bit is clear keeps the lower half and
// receives the partner's lower half, the other lane the upper half), so after them every lane carries one quantity;
// the remaining steps are the plain butterfly.  Each addition is still  own + partner's  at the same offset, i.e.
// bit-for-bit the additions of wave_butterfly.  Result valid in every thread.
template <int NQ, int NW>
__device__ __forceinline__ void block_reduce_pow2(double (&v)[NQ], double* lds) {
    static_assert(NQ == 2 || NQ == 4 || NQ == 8, "power of two");
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int q = 0;                                   // quantity carried in slot 0 once the halving is over
    int off = 32;
#pragma unroll
    for (int h = NQ / 2; h >= 1; h >>= 1, off >>= 1) {
        const bool up = (lane & off) != 0;
#pragma unroll
        for (int i = 0; i < h; ++i) {
            const double keep = up ? v[i + h] : v[i];
            const double send = up ? v[i] : v[i + h];
            v[i] = keep + __shfl_xor(send, off, 64);
        }
        if (up) q += h;
    }
    double t = v[0];
    for (; off >= 1; off >>= 1) t = t + __shfl_xor(t, off, 64);
    // lanes whose low bits (below the halving bits) are zero publish their quantity
    constexpr int LOW = 64 / NQ;                 // 32 / 16 / 8 lanes share one quantity
    if ((lane & (LOW - 1)) == 0) lds[q * NW + wave] = t;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < NQ; ++k) {
        double s = lds[k * NW];
#pragma unroll
        for (int w = 1; w < NW; ++w) s = s + lds[k * NW + w];
        v[k] = s;
    }
    __syncthreads();
}
template <int NQ, int NW>
__device__ __forceinline__ void block_reduce_any(double (&v)[NQ], double* lds) {
    if constexpr (NQ == 2 || NQ == 4 || NQ == 8) block_reduce_pow2<NQ, NW>(v, lds);
    else block_reduce<NQ, NW>(v, lds);
}

// Two-level fold of NQ arrays of tile partials (one launch, gridDim.x = nchunks = ceil(ntiles / KR_F) workgroups of
// KR_F threads):
//   stage 1  workgroup c folds partials [c*KR_F, (c+1)*KR_F): thread t takes partial c*KR_F + t (0.0 past the end),
//            64-lane butterfly, serial fold over the 16 waves -> chunk value c;
//   stage 2  (only when nchunks > 1) the workgroup that finishes LAST folds the chunk values: thread t folds chunks
//            t, t+KR_F, ... in ascending order, butterfly, serial over waves.
// The association tree depends only on ntiles, never on which workgroup happens to be last.  Hand-off: write-through
// (agent-scope) stores of the chunk values, waited for, then the ticket atomic; the last workgroup reads them with
// agent-scope loads (cdna_hip_programming.md, Guideline 16, without the L2-wide fences).  Returns true in the workgroup that holds
// the final result (valid in every thread of it).
// Round 3, the hand-off without the ticket (ticket == nullptr; KRYST_FOLD_POLL=0 keeps the ticket): a chunk value is its own flag -- the
// chunk cells hold a NaN payload no fold produces until a workgroup stores its value there (write-through, NOT waited for), and
// workgroup 0 polls the cells it is going to fold anyway (agent-scope loads) and puts the payload back behind itself for the next
// fold.  Two dependent round trips less than store -> acknowledge -> ticket atomic -> load (5.8 -> ~4 us per fold at 256^3, where an
// iteration has two to thirty-one of them).  Both forms leave the cells armed, so they can alternate.
#define KR_FOLD_UNSET 0x7FF8F01DF01DF01Dull
// A polling hand-off whose patience runs out (seconds: the GPU is shared, time-sliced or serialised by a profiler) raises *err (it stays
// raised) instead of folding the payload as if it were a value: the consumers end the solve with KRYST_ERR_HIP, and the host switches the
// context to the ticket form, whose cells are written before they are read (a late store of the abandoned fold cannot be mistaken for a
// value there).
template <int NQ>
__device__ __forceinline__ int fold2(const double* partials, int64_t stride, int64_t ntiles, double* chunks,
                                     int64_t cstride, unsigned int* ticket, unsigned int* err, double (&out)[NQ], double* lds) {
    // returns 0 in a workgroup that does not hold the result, 1 in the one that does (valid in every thread), 2 there when the polling
    // hand-off gave up (no result; *err is raised)
    __shared__ int is_last, gave_up;
    const int64_t i = (int64_t)blockIdx.x * KR_F + threadIdx.x;
    if (threadIdx.x == 0) gave_up = 0;
#pragma unroll
    for (int q = 0; q < NQ; ++q) out[q] = (i < ntiles) ? partials[q * stride + i] : 0.0;
    block_reduce_any<NQ, KR_F / 64>(out, lds);
    if (gridDim.x == 1) return 1;
    const double unset = __longlong_as_double((long long)KR_FOLD_UNSET);
    if (ticket == nullptr) {
        if (threadIdx.x == 0) {
#pragma unroll
            for (int q = 0; q < NQ; ++q) __hip_atomic_store(&chunks[q * cstride + blockIdx.x], out[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (blockIdx.x != 0) return 0;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            double acc = 0.0;
            for (int64_t j = threadIdx.x; j < (int64_t)gridDim.x; j += KR_F) {
                double v = __hip_atomic_load(&chunks[q * cstride + j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                // (the other workgroups may still be waiting for a free slot behind another stream's kernel: patience of seconds, short naps
                // first)
                for (int budget = 1 << 22; (unsigned long long)__double_as_longlong(v) == KR_FOLD_UNSET && budget > 0; --budget) {
                    if (budget > (1 << 22) - 4096) __builtin_amdgcn_s_sleep(1); else __builtin_amdgcn_s_sleep(64);
                    v = __hip_atomic_load(&chunks[q * cstride + j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                if ((unsigned long long)__double_as_longlong(v) == KR_FOLD_UNSET) {      // never delivered: no value, and nothing to arm again
                    gave_up = 1;
                    __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    continue;
                }
                acc = acc + v;
                chunks[q * cstride + j] = unset;                  // armed again for the next (stream-ordered) fold
            }
            out[q] = acc;
        }
        block_reduce_any<NQ, KR_F / 64>(out, lds);               // (its barriers also order the gave_up stores in front of the read below)
        return gave_up ? 2 : 1;
    }
    if (threadIdx.x == 0) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) __hip_atomic_store(&chunks[q * cstride + blockIdx.x], out[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // the chunk values are write-through (agent-scope) stores: once they have been acknowledged they are visible to the agent-scope
        // loads of the last workgroup.  (An agent-scope FENCE writes back / invalidates the whole L2 on this chip -- measured in
        // ilu.hip's factorisation kernel: 13 per row made a 14 ms launch 190 ms -- and is not needed on either side.)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned int t = atomicAdd(ticket, 1u);
        is_last = (t == gridDim.x - 1) ? 1 : 0;
    }
    __syncthreads();
    if (!is_last) return 0;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        double acc = 0.0;
        for (int64_t j = threadIdx.x; j < (int64_t)gridDim.x; j += KR_F) {
            acc = acc + __hip_atomic_load(&chunks[q * cstride + j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            chunks[q * cstride + j] = unset;                      // (the cells stay armed for the polling form)
        }
        out[q] = acc;
    }
    block_reduce_any<NQ, KR_F / 64>(out, lds);
    if (threadIdx.x == 0) *ticket = 0u;            // ready for the next (stream-ordered) fold
    return 1;
}
#endif

// ---- launchers implemented in blas1.hip / spmv.hip / precond.hip (all enqueue on ctx->s_main) ----
// the ticket of the two-level fold, or nullptr for the polling hand-off (fold2; KRYST_FOLD_POLL, read per launch)
// (after a polling hand-off has timed out once -- fold_err, below -- the context keeps the ticket form: its cells are written before they are read)
inline unsigned int* fold_ticket(kryst_ctx_t ctx) { return (ctx->fold_poll_off || env_int("KRYST_FOLD_POLL", 1) == 0) ? ctx->d_ticket : nullptr; }
// the error word of the polling hand-off: set by the device when workgroup 0's patience ran out (a time-sliced or profiled GPU)
inline unsigned int* fold_err(kryst_ctx_t ctx) { return ctx->d_ticket + 8; }
int32_t launch_dot_partials(kryst_ctx_t ctx, const double* x, const double* y, int64_t n, int slot);
// local result of up to nq partial arrays -> d_out[0..nq) (device), single rank: the final value
int32_t launch_final_fold(kryst_ctx_t ctx, int nq, int64_t ntiles, double* d_out);
// reads and clears the polling hand-off's error word; when it was raised: sets the error text, switches the context to the ticket form
// and returns true (blas1.hip)
bool fold_gave_up(kryst_ctx_t ctx);
int32_t vec_check2(kryst_vec_t a, kryst_vec_t b);
bool use_collectives(kryst_ctx_t ctx);

}  // namespace kr
