// The begin / iterate / end skeleton every non-restarted solver shares, plus the kernels and logic steps that handle
// an exit whose last vector update is still pending (`early`).
#pragma once
#include "solver_common.h"

namespace kr {

struct ClearEarlyLogic {
    static constexpr bool RUN_WHEN_DONE = true;
    LogicCtx c;
    __device__ void run(const double*) const { c.st->early = 0; }
};
// gates for exits whose last vector update is still pending (`early`)
struct GateEarly {                   // runs until the solve has ended, and once more while `early` is raised
    const DevState* st;
    __device__ __forceinline__ bool skip() const { return st->done && !st->early; }
};
struct GateIfEarly {                 // runs only while `early` is raised
    const DevState* st;
    __device__ __forceinline__ bool skip() const { return !st->early; }
};

inline int32_t solve_args_check(const SolveIO& io, kryst_vec_t b, kryst_vec_t x) {
    KR_ARG(io.a && io.params && b && x, "solve: null argument");
    KR_ARG(b->ctx == io.a->ctx && x->ctx == io.a->ctx, "solve: context mismatch");
    KR_ARG(io.a->nrows == io.a->xlen, "solve: square operator required");
    KR_ARG(b->n == io.a->nrows && x->n == io.a->nrows, "solve: vector length != operator size");
    KR_ARG(io.params->max_iters >= 0, "solve: max_iters < 0");
    KR_ARG(!io.pc || io.pc->ctx == io.a->ctx, "solve: preconditioner belongs to another context");
    KR_ARG(!io.pc || io.pc->n < 0 || io.pc->n == io.a->nrows, "solve: preconditioner size mismatch");
    return KRYST_OK;
}

// the callbacks the poll loop has not fired yet (the stream is idle: every entry is in host memory)
inline void finish_monitor(LiveMonitor& mon, Workspace& ws) {
    if (!mon.io || !mon.io->monitor) return;
    DevState h;
    if (read_state(ws, &h) != hipSuccess) return;
    mon.upto(h.hist_len);
}

// A solve split into begin / iterate / end so that the same code serves the one-shot LinearSolver::solve
// entry points and the stepping session bench.py uses to time exactly K iterations.
struct SolverRun {
    kryst_vec_t bv, xv; SolveIO io; kryst_params_t prm;
    kryst_csr_t a; kryst_ctx_t ctx; int64_t n, nt;
    Workspace ws; LogicCtx lc; const int* done = nullptr; double* xw = nullptr;
    kryst_pc_s pcl; kryst_pc_t pc = nullptr;
    int64_t next_iter = 1;
    LiveMonitor mon;
    SolverRun(kryst_vec_t b, kryst_vec_t x, const SolveIO& io_)
        : bv(b), xv(x), io(io_), prm(*io_.params), a(io_.a), ctx(io_.a->ctx), n(io_.a->nrows), nt(ntiles_of(io_.a->nrows)),
          ws(io_.a->ctx, io_.a->nrows) { io.params = &prm; }
    virtual ~SolverRun() { if (a) a->halo_started_for = nullptr; }
    virtual int32_t begin() = 0;
    virtual int32_t iterate(int64_t i) = 0;
    virtual int32_t flush() { return KRYST_OK; }     // work a solver still owes once its last iteration has been enqueued (CG / PCG: the deferred x update)
    int32_t common_begin(int64_t hist_entries, int work_vectors) {
        KR_HIP(hipSetDevice(ctx->device));
        a->halo_started_for = nullptr;        // (an early halo start belongs to ONE solve: work vectors of later solves reuse the addresses)
        if (io.pc) { pcl = *io.pc; if (pcl.n < 0) pcl.n = n; pc = &pcl; }
        KR_TRY(ws.init(hist_entries));
        KR_TRY(ws.reserve(work_vectors + 1));
        lc = ws.lctx(&prm, io.monitor != nullptr);
        mon.io = &io; mon.ws = &ws; mon.first = 0;
        done = &ws.st->done;
        KR_TRY(ws.vec(&xw));
        KR_HIP(hipMemcpyAsync(xw, xv->d, padded_bytes(n), hipMemcpyDeviceToDevice, ctx->s_main));
        return KRYST_OK;
    }
    int32_t step(int64_t k) {                // enqueue k more iterations, no host synchronisation
        const EnvFreeze knobs;               // the tuning knobs are read once per step call, not per launch
        for (int64_t j = 0; j < k && next_iter <= prm.max_iters; ++j, ++next_iter) KR_TRY(iterate(next_iter));
        return KRYST_OK;
    }
    int32_t end() {
        KR_TRY(flush());
        KR_HIP(hipStreamSynchronize(ctx->s_comm));
        KR_HIP(hipStreamSynchronize(ctx->s_main));
        a->halo_started_for = nullptr;
        const int32_t status = finish_solve(ws, io);
        if (status == KRYST_OK)                  // on Err the reference never reaches `*x = ...`
            KR_HIP(hipMemcpyAsync(xv->d, xw, padded_bytes(n), hipMemcpyDeviceToDevice, ctx->s_main));
        KR_HIP(hipStreamSynchronize(ctx->s_main));
        finish_monitor(mon, ws);
        return status;
    }
    int32_t solve() {
        const EnvFreeze knobs;               // the tuning knobs are read once per solve, not per launch
        KR_TRY(begin());
        KR_TRY(run_ahead(ctx, &prm, [&](int64_t i) -> int32_t { next_iter = i + 1; return iterate(i); }, [&] { mon.poll(); }));
        return end();
    }
};

}  // namespace kr
