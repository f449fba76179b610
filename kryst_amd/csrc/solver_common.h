// Device-resident Krylov iteration: shared state, scalar "logic" kernels and the host enqueue/poll loop.
//
// The reference solvers interleave vector loops with scalar decisions on the host (alpha = rz/pAp, the
// convergence test, breakdown checks).  Here the scalars never leave the GPU: every inner product ends in a
// one-workgroup kernel that folds the tile partials (fixed order) and then runs the solver's scalar step in
// thread 0, writing alpha/beta/... to device memory where the next vector kernel reads them.  When the
// reference would return, the logic sets `done`; every later kernel of the stream starts with `if (*done)
// return`, so the host can enqueue iterations ahead without a sync per iteration and the result is still
// EXACTLY the reference's iterate (same iteration count, same x).  The host learns about `done` from a
// progress record in mapped host memory.
#pragma once
#include "pc.h"
#include "ew.h"
#include <cfloat>

namespace kr {

// A vector that the NEXT launch reads again is written / read with cacheable accesses when it can survive in the 256 MB
// memory-side cache, and streamed otherwise (measured: r of CG kept, 128^3 +5 %, 256^3 +1 %, 512^3 -5 %; the Gram-Schmidt
// links' z kept, 128^3 / 256^3 +8 %).  KRYST_KEEP_BYTES moves the threshold (0: never keep).
inline bool keep_in_cache(int64_t n) {
    const long long lim = env_ll("KRYST_KEEP_BYTES", 160ll << 20);   // (tools/solver_ab.py changes it between session steps)
    return (long long)n * 8 <= lim;
}


struct DevState {
    // generic scalars
    double rsq, alpha, beta, res0, rz, normq, rho, rho_prev, omega, omega_prev;
    double final_residual;
    long long iter;          // completed iterations
    long long iterations;    // SolveStats.iterations
    long long hist_len;
    int done;                // the reference would have returned / broken out
    int status;              // KRYST_* code
    int converged;           // SolveStats.converged
    int early;               // BiCGStab: s-norm exit pending its x update
    long long xlast;         // CG / PCG with the deferred x update: the iteration that ended the solve AFTER the reference's x += alpha p
                             // (its direction pass still has to add alpha p to x; iterations count from 1, 0 = none)
    long long xpend;         // CG / PCG with the direction pass inside the SpMV: the iteration whose x += alpha p is still owed (the fused
                             // SpMV of iteration xpend + 1 pays it, or the flush at the end of the solve / session)
    double alpha_hist[16];   // ... with x updated in BATCHES (solvers.hip: XBatchOp): alpha of iteration i at [i % ring]
};

struct LogicCtx {
    DevState* st; double* hist; HostProgress* prog; const double* red;
    double tol; long long max_iters; int norm_type;
    long long hist_cap; int live;        // entries the mapped history buffer holds; live: a monitor is attached, publish the count
    // residual_history.push (cg.rs:140,263): the count always advances, entries beyond the buffer are not recorded
    __device__ __forceinline__ void push(double v) const {
        const long long k = st->hist_len;
        if (k < hist_cap) hist[k] = v;
        st->hist_len = k + 1;
        if (live) { __threadfence_system(); prog->hist_len = k + 1; }
    }
    __device__ __forceinline__ void finish(int status) const {
        st->status = status; st->done = 1;
        prog->iter = st->iter + 1; prog->res = st->final_residual; prog->status = status;
        __threadfence_system();
        prog->done = 1;
        __threadfence_system();
    }
    // Convergence::check (src/utils/convergence.rs:18-34)
    __device__ __forceinline__ bool check(double res, double res0, long long i) const {
        const double rel = res / res0;
        const bool conv = (rel <= tol) || (i >= max_iters);
        st->iterations = i; st->final_residual = res; st->converged = conv ? 1 : 0;
        return conv;
    }
};

// correctly rounded fp64 sqrt on the device (every residual-history entry of the bit-exact solver tests goes through it)
__device__ __forceinline__ double dsqrt(double x) { return __builtin_sqrt(x); }

// single rank: fold the tile partials and run the logic in one launch
template <int NQ, class L>
__global__ __launch_bounds__(KR_F) void fold_logic_kernel(const double* partials, int64_t stride, int64_t ntiles,
                                                          double* chunks, int64_t cstride, unsigned int* ticket, unsigned int* err,
                                                          double* red_out, L logic) {
    if (logic.c.st->done && !L::RUN_WHEN_DONE) return;      // uniform over the grid: only the LAST workgroup ever sets done
    __shared__ double lds[NQ * (KR_F / 64)];
    double v[NQ];
    const int f = fold2<NQ>(partials, stride, ntiles, chunks, cstride, ticket, err, v, lds);
    if (!f) return;
    if (threadIdx.x == 0) {
        if (f == 2) { logic.c.finish(KRYST_ERR_HIP); return; }      // the hand-off gave up: no value to act on (finish_solve explains)
#pragma unroll
        for (int q = 0; q < NQ; ++q) red_out[q] = v[q];
        logic.run(red_out);
    }
}
template <class L>
__global__ void logic_kernel(const double* red, L logic) {
    if (logic.c.st->done && !L::RUN_WHEN_DONE) return;
    if (threadIdx.x == 0) logic.run(red);
}
// several ranks: fold the all-gathered rank results in rank order (total = r0; total = total + r_p), then the logic
template <int NQ, class L>
__global__ void rank_fold_logic_kernel(const double* gathered, int nranks, const unsigned int* err, double* red_out, L logic) {
    if (logic.c.st->done && !L::RUN_WHEN_DONE) return;
    if (threadIdx.x != 0) return;
    if (*err) { logic.c.finish(KRYST_ERR_HIP); return; }       // this rank's local fold gave up (its NaN went to the peers)
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        double total = gathered[q];
        for (int p = 1; p < nranks; ++p) total = total + gathered[p * NQ + q];
        red_out[q] = total;
    }
    logic.run(red_out);
}

// several ranks, mailbox path (kryst_ctx_s::ipc_*): local two-level fold, exchange and rank-ordered fold + logic in ONE launch.
// The workgroup that ends up with the local result sends it -- lane p -> rank p's mailbox cell [epoch parity][my rank] -- polls its own
// mailbox for rank p's message of this epoch, and folds the P values in rank order (total = r0; total = total + r_p): the bits of the
// all-gather path.  Round 4: the cells are SELF-VALIDATING -- a value travels as two 8-byte words, (low half | tag) and (high half | tag),
// tag = the epoch's low 31 bits with the top bit set, each word one system-scope store that is atomic on its own; the receiver takes a
// value when both tags match.  No acknowledge wait between data and stamp, no stamp: one round trip less per inner product than the
// round-3 form (values, s_waitcnt vmcnt(0), stamp).  Two cells per writer (epoch parity) suffice: a rank can be at most one reduction
// ahead of the slowest one, because finishing reduction e needs everybody's contribution to e.  A peer that never shows up (budget) ends
// the solve with KRYST_ERR_RCCL instead of a hung GPU.
struct IpcView { double* mine; double* const* peers; unsigned long long* epoch; int me, P, budget; };
template <int NQ, class L>
__global__ __launch_bounds__(KR_F) void fold_ipc_logic_kernel(const double* partials, int64_t stride, int64_t ntiles,
                                                              double* chunks, int64_t cstride, unsigned int* ticket, unsigned int* err,
                                                              double* red_out, L logic, IpcView v) {
    if (logic.c.st->done && !L::RUN_WHEN_DONE) return;      // the same decision on every rank (identical scalars): nobody sends, nobody waits
    __shared__ double lds[NQ * (KR_F / 64)];
    double val[NQ];
    const int f = fold2<NQ>(partials, stride, ntiles, chunks, cstride, ticket, err, val, lds);
    if (!f) return;
    if (threadIdx.x >= 64) return;
    if (f == 2) {                                                    // the local hand-off gave up: the peers still get their message (a NaN)
#pragma unroll
        for (int q = 0; q < NQ; ++q) val[q] = __longlong_as_double(0x7FF8000000000000ll);
    }
    const int p = threadIdx.x;
    const unsigned long long e = *v.epoch + 1;
    const int par = (int)(e & 1ull);
    const unsigned long long tag = ((e & 0x7fffffffull) | 0x80000000ull) << 32;
    if (p < v.P) {
        unsigned long long* dst = reinterpret_cast<unsigned long long*>(v.peers[p] + (size_t)(par * v.P + v.me) * 16);
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const unsigned long long bits = (unsigned long long)__double_as_longlong(val[q]);
            __hip_atomic_store(dst + 2 * q, (bits & 0xffffffffull) | tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(dst + 2 * q + 1, (bits >> 32) | tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    bool ok = true;
    double got[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) got[q] = 0.0;
    if (p < v.P) {
        const unsigned long long* src = reinterpret_cast<const unsigned long long*>(v.mine + (size_t)(par * v.P + p) * 16);
        int b = v.budget;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            unsigned long long w0 = __hip_atomic_load(src + 2 * q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            unsigned long long w1 = __hip_atomic_load(src + 2 * q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            while (((w0 ^ tag) >> 32) != 0ull || ((w1 ^ tag) >> 32) != 0ull) {
                if (--b <= 0) { ok = false; break; }
                __builtin_amdgcn_s_sleep(2);
                w0 = __hip_atomic_load(src + 2 * q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                w1 = __hip_atomic_load(src + 2 * q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
            got[q] = __longlong_as_double((long long)((w0 & 0xffffffffull) | (w1 << 32)));
        }
    }
    const bool all_ok = __all(ok);
    // rank-ordered fold: lane r holds rank r's values
    double total[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        total[q] = __shfl(got[q], 0, 64);
        for (int r = 1; r < v.P; ++r) total[q] = total[q] + __shfl(got[q], r, 64);
    }
    if (p != 0) return;
    *v.epoch = e;
    if (!all_ok) { logic.c.finish(KRYST_ERR_RCCL); return; }
    if (f == 2) { logic.c.finish(KRYST_ERR_HIP); return; }
#pragma unroll
    for (int q = 0; q < NQ; ++q) red_out[q] = total[q];
    logic.run(red_out);
}

template <int NQ, class L>
inline int32_t reduce_then(kryst_ctx_t ctx, int64_t ntiles, double* d_red, const L& logic) {
    if (!use_collectives(ctx)) {
        KR_TRY(ensure_partials(ctx, ntiles > 0 ? ntiles : 1));
        hipLaunchKernelGGL((fold_logic_kernel<NQ, L>), dim3((unsigned)nchunks_of(ntiles)), dim3(KR_F), 0, ctx->s_main,
                           ctx->d_partials, ctx->partials_cap, ntiles, ctx->d_chunks, ctx->chunks_cap, fold_ticket(ctx), fold_err(ctx), d_red, logic);
    } else if (ctx->ipc_on) {
        KR_TRY(ensure_partials(ctx, ntiles > 0 ? ntiles : 1));
        static const int budget = [] { const char* e = getenv("KRYST_IPC_POLL_BUDGET"); return e ? std::max(1, atoi(e)) : (1 << 26); }();
        const IpcView v{ctx->ipc_mine, ctx->d_ipc_peers, ctx->d_ipc_epoch, ctx->rank, ctx->nranks, budget};
        hipLaunchKernelGGL((fold_ipc_logic_kernel<NQ, L>), dim3((unsigned)nchunks_of(ntiles)), dim3(KR_F), 0, ctx->s_main,
                           ctx->d_partials, ctx->partials_cap, ntiles, ctx->d_chunks, ctx->chunks_cap, fold_ticket(ctx), fold_err(ctx), d_red, logic, v);
    } else {
        // local two-level fold -> RCCL all-gather of NQ doubles per rank -> rank-ordered fold + logic in one launch
        double* local = ctx->d_gather + (size_t)ctx->nranks * KR_MAXQ;
        KR_TRY(launch_final_fold(ctx, NQ, ntiles, local));
        KR_TRY(comm_all_gather(ctx, local, ctx->d_gather, NQ));
        hipLaunchKernelGGL((rank_fold_logic_kernel<NQ, L>), dim3(1), dim3(64), 0, ctx->s_main, ctx->d_gather, ctx->nranks, fold_err(ctx), d_red, logic);
    }
    KR_HIP(hipGetLastError());
    phase_mark(ctx, KR_PH_REDUCE);
    return KRYST_OK;
}

// Work area of one solve: device vectors, state, mapped history.  Freed by the destructor.
struct Workspace {
    kryst_ctx_t ctx; int64_t n;
    std::vector<double*> vecs;
    DevState* st = nullptr; double* red = nullptr;
    double* h_hist = nullptr; double* d_hist = nullptr; int64_t hist_cap = 0;
    bool owns_arena = false;
    explicit Workspace(kryst_ctx_t c, int64_t n_) : ctx(c), n(n_) {}
    ~Workspace() {
        if (ctx->active_ws == this) ctx->active_ws = nullptr;
        (void)hipStreamSynchronize(ctx->s_main);
        for (double* p : vecs) (void)hipFree(p);
        if (h_hist) (void)hipHostFree(h_hist);
        if (owns_arena) { ctx->arena_owner = nullptr; ctx->arena_used = 0; }
    }
    size_t vec_bytes() const {
        const size_t b = sizeof(double) * (size_t)((n + KR_TILE - 1) / KR_TILE * KR_TILE + KR_TILE);
        return (b + 4095) & ~(size_t)4095;
    }
    // Claim the context's arena for `count` work vectors (grown if needed).  A second solve that starts while another
    // workspace of the same context is alive (a stepping session) falls back to one allocation per vector.
    int32_t reserve(int64_t count) {
        if (ctx->arena_owner != nullptr) return KRYST_OK;
        static const bool off = [] { const char* e = getenv("KRYST_NO_ARENA"); return e && atoi(e) != 0; }();   // measurement knob
        if (off) return KRYST_OK;
        const size_t need = vec_bytes() * (size_t)count;
        if (need > ctx->arena_bytes) {
            KR_HIP(hipStreamSynchronize(ctx->s_main));
            if (ctx->arena) { KR_HIP(hipFree(ctx->arena)); ctx->arena = nullptr; ctx->arena_bytes = 0; }
            if (hipMalloc(&ctx->arena, need) != hipSuccess) { (void)hipGetLastError(); return KRYST_OK; }   // per-vector fallback
            ctx->arena_bytes = need;
        }
        ctx->arena_owner = this; ctx->arena_used = 0; owns_arena = true;
        return KRYST_OK;
    }
    int32_t vec(double** out) {
        const size_t bytes = vec_bytes();
        double* p = nullptr;
        if (owns_arena && ctx->arena_used + bytes <= ctx->arena_bytes) {
            p = reinterpret_cast<double*>(ctx->arena + ctx->arena_used);
            ctx->arena_used += bytes;
        } else {
            KR_HIP(hipMalloc(&p, bytes));
            vecs.push_back(p);
        }
        KR_HIP(hipMemsetAsync(p, 0, bytes, ctx->s_main));
        *out = p;
        return KRYST_OK;
    }
    // history entries kept per solve: max_iters + 2, but never more than this (a caller's "effectively unbounded"
    // max_iters must not turn into gigabytes of pinned memory); the reported length still counts every push
    static constexpr int64_t HIST_MAX = (int64_t)1 << 22;
    int32_t init(int64_t hist_entries) {
        if (ctx->active_ws != nullptr) {
            set_error("context busy: a solve or stepping session is already open on this context (one at a time)");
            return KRYST_ERR_BUSY;
        }
        ctx->active_ws = this;
        hist_entries = std::max<int64_t>(2, std::min<int64_t>(hist_entries, HIST_MAX));
        st = reinterpret_cast<DevState*>(ctx->d_scal);
        red = ctx->d_scal + 256;
        KR_HIP(hipMemsetAsync(ctx->d_scal, 0, sizeof(double) * 512, ctx->s_main));
        hist_cap = hist_entries;
        KR_HIP(hipHostMalloc((void**)&h_hist, sizeof(double) * (size_t)hist_cap, hipHostMallocMapped));
        KR_HIP(hipHostGetDevicePointer((void**)&d_hist, h_hist, 0));
        ctx->h_prog->iter = 0; ctx->h_prog->res = 0; ctx->h_prog->done = 0; ctx->h_prog->status = 0; ctx->h_prog->hist_len = 0;
        return KRYST_OK;
    }
    LogicCtx lctx(const kryst_params_t* p, bool live = false) const {
        return LogicCtx{st, d_hist, ctx->d_prog, red, p->tol, (long long)p->max_iters, p->norm_type, (long long)hist_cap, live ? 1 : 0};
    }
};

// the solve's scalar state, read in stream order on the compute stream (the product issues nothing on the null stream)
inline hipError_t read_state(const Workspace& ws, DevState* h) {
    const hipError_t e = hipMemcpyAsync(h, ws.st, sizeof(DevState), hipMemcpyDeviceToHost, ws.ctx->s_main);
    return e != hipSuccess ? e : hipStreamSynchronize(ws.ctx->s_main);
}

inline size_t padded_bytes(int64_t n) { return sizeof(double) * (size_t)((n + KR_TILE - 1) / KR_TILE * KR_TILE); }

// Host side of the run-ahead loop: enqueue `body(i)` for i = 1..max_iters in batches, stop once the device has
// published `done`.  With several ranks every rank must enqueue the same collectives, so a rank only acts on a
// `done` that happened inside batches it has fully waited for (all ranks then see the same thing).
struct NoPoll { void operator()() const {} };
template <class Body, class Poll = NoPoll>
inline int32_t run_ahead(kryst_ctx_t ctx, const kryst_params_t* p, Body body, Poll poll = Poll()) {
    const int64_t chk = p->check_every > 0 ? p->check_every : 8;
    int64_t it = 0, batch = 0;
    int64_t synced_iters = 0;
    std::vector<int64_t> batch_end;
    while (it < p->max_iters) {
        const int64_t nb = std::min<int64_t>(chk, p->max_iters - it);
        for (int64_t k = 0; k < nb; ++k) KR_TRY(body(it + k + 1));
        it += nb;
        batch_end.push_back(it);
        KR_HIP(hipEventRecord(ctx->ev_ring[batch & 3], ctx->s_main));
        if (batch >= 1) {                                   // at most two batches in flight
            KR_HIP(hipEventSynchronize(ctx->ev_ring[(batch - 1) & 3]));
            synced_iters = batch_end[(size_t)batch - 1];
        }
        poll();                                             // live monitor: report what the device has pushed so far
        ++batch;
        if (ctx->h_prog->done && (ctx->nranks == 1 || ctx->h_prog->iter <= synced_iters)) break;
    }
    KR_HIP(hipStreamSynchronize(ctx->s_comm));
    KR_HIP(hipStreamSynchronize(ctx->s_main));
    return KRYST_OK;
}

struct DotOneOp {
    static constexpr int NQ = 1; static constexpr const char* TAG = "DotOne"; static constexpr int BPC = 4;
    const double *a, *b;
    __device__ __forceinline__ void pair(int64_t i, bool in0, bool in1, double (&acc)[1]) const {
        const d2 u = ld2(a, i), v = ld2(b, i);
        if (in0) acc[0] = acc[0] + u.a * v.a;
        if (in1) acc[0] = acc[0] + u.b * v.b;
    }
};
// out = a - b, partial of out.out        (r = b - A x, `bi - ax`, cg.rs:123; ||r||^2 for the first dot)
struct SubDotOp {
    static constexpr int NQ = 1; static constexpr const char* TAG = "SubDot";
    const double *a, *b; double* out;
    __device__ __forceinline__ void pair(int64_t i, bool in0, bool in1, double (&acc)[1]) const {
        const d2 u = ld2(a, i), v = ld2(b, i);
        const double r0 = u.a - v.a, r1 = u.b - v.b;
        st2(out, i, r0, r1);
        if (in0) acc[0] = acc[0] + r0 * r0;
        if (in1) acc[0] = acc[0] + r1 * r1;
    }
};
// r = b - A x with ||r||^2 partials; tmp receives A x
inline int32_t residual_dot(kryst_csr_t a, const double* b, const double* x, double* r, double* tmp, const int* done) {
    KR_TRY(launch_spmv(a, x, tmp, 0, nullptr, done));
    return launch_ew(a->ctx, SubDotOp{b, tmp, r}, a->nrows, done);
}

struct SolveIO {
    kryst_csr_t a; kryst_pc_t pc; const kryst_params_t* params; kryst_stats_t* stats;
    double* hist; int64_t hist_cap; int64_t* hist_len; kryst_monitor_fn monitor; void* user;
};

// with_monitor (cg.rs:84-88): the reference calls the monitor inside the loop (cg.rs:137-140,260-263; pcg.rs:143-146,196-199).
// Here the device pushes every history entry into mapped host memory and publishes the count; the host fires the callbacks
// on the calling thread, in order, each time its poll loop has waited for a batch (every `check_every` iterations) and
// once more when the solve has ended -- always before the solve call returns.
struct LiveMonitor {
    const SolveIO* io = nullptr; Workspace* ws = nullptr; int64_t first = 0; int64_t reported = 0;
    void upto(int64_t len) {
        if (!io || !io->monitor) return;
        len = std::min<int64_t>(len, ws->hist_cap);
        for (; reported < len; ++reported) io->monitor(first + reported, ws->h_hist[reported], io->user);
    }
    void poll() { if (io && io->monitor) upto(ws->ctx->h_prog->hist_len); }
};

// copy stats / history out and run the monitor callbacks (in order, on the calling thread)
inline int32_t finish_solve(Workspace& ws, const SolveIO& io) {
    kryst_ctx_t ctx = ws.ctx;
    DevState h;
    KR_HIP(read_state(ws, &h));
    if (io.stats) { io.stats->iterations = h.iterations; io.stats->final_residual = h.final_residual; io.stats->converged = h.converged; }
    const int64_t len = h.hist_len;
    if (io.hist_len) *io.hist_len = len;
    for (int64_t k = 0; k < len && k < ws.hist_cap; ++k) {
        if (io.hist && k < io.hist_cap) io.hist[k] = ws.h_hist[k];
    }
    // a preconditioner apply abandoned by the device (pc.h: pc_health) voids the whole solve, whatever the recurrences made of it
    if (io.pc && pc_health(io.pc) != KRYST_OK) return KRYST_SOLVE_ERROR;
    // a fold's polling hand-off (status raised by the device) or a halo pull (NaNs in the halo, no status) that ran out of patience
    if ((h.status == KRYST_ERR_HIP || (ctx->nranks > 1 && io.a->plan.peer.on)) && fold_gave_up(ctx)) return h.status != KRYST_OK ? h.status : KRYST_ERR_RCCL;
    return h.status;
}

}  // namespace kr
