// Device-resident CG / PCG / BiCGStab / GMRES(m): LinearSolver::solve (src/solver/mod.rs:30-52) for a HIP CSR
// operator.  Each solver restates its reference file operation by operation (line numbers cited inline); the
// vector work is fused into as few HBM passes as the data dependences allow without changing any rounding.
#include "solver_run.h"

namespace kr {

// =================================================================== shared vector ops
struct DotPairOp {
    static constexpr int NQ = 2; static constexpr const char* TAG = "DotPair";
    const double *a, *b, *c, *d;
    __device__ __forceinline__ void pair(int64_t i, bool in0, bool in1, double (&acc)[2]) const {
        const d2 u = ld2(a, i), v = ld2(b, i), w = ld2(c, i), z = ld2(d, i);
        if (in0) { acc[0] = acc[0] + u.a * v.a; acc[1] = acc[1] + w.a * z.a; }
        if (in1) { acc[0] = acc[0] + u.b * v.b; acc[1] = acc[1] + w.b * z.b; }
    }
};
template <bool KEEP = false>
struct AypxDevOp {                   // y = x + beta*y  (cg.rs:274-276, pcg.rs:215-217), beta on the device
    static constexpr int NQ = 0; static constexpr const char* TAG = "AypxDev"; static constexpr int PHASE = KR_PH_BLAS1_DIRECTION;
    const double* beta; const double* x; double* y;
    __device__ __forceinline__ void pair(int64_t i, bool, bool, double (&)[1]) const {
        const double be = *beta;
        const d2 a = ld2_sel<KEEP>(x, i), b = ld2(y, i);                 // x (the residual) was written by the launch before this one
        st2(y, i, a.a + be * b.a, a.b + be * b.b);
    }
};

// =================================================================== CG (src/solver/cg.rs:114-288)
// x += alpha p ; r -= alpha Ap (cg.rs:207-212) ; partial r.r (cg.rs:223) [; partial r.p for the Natural norm, :227]
template <bool KEEP = false>
struct CgUpdate1 {
    static constexpr int NQ = 1; static constexpr const char* TAG = "CgUpdate1"; static constexpr int PHASE = KR_PH_BLAS1_RESIDUAL;
    const double* alpha; const double* p; const double* ap; double* x; double* r;
    __device__ __forceinline__ void pair(int64_t i, bool in0, bool in1, double (&acc)[1]) const {
        const double al = *alpha;
        const d2 pp = ld2(p, i), aa = ld2(ap, i), xx = ld2(x, i), rr = ld2(r, i);
        const double x0 = xx.a + al * pp.a, x1 = xx.b + al * pp.b;
        const double r0 = rr.a - al * aa.a, r1 = rr.b - al * aa.b;
        st2(x, i, x0, x1); st2_sel<KEEP>(r, i, r0, r1);                   // r is read again by the next launch (p = r + beta p)
        if (in0) acc[0] = acc[0] + r0 * r0;
        if (in1) acc[0] = acc[0] + r1 * r1;
    }
};
struct CgUpdate2 {                   // + r.p (old p) for CgNormType::Natural
    static constexpr int NQ = 2; static constexpr const char* TAG = "CgUpdate2";
    const double* alpha; const double* p; const double* ap; double* x; double* r;
    __device__ __forceinline__ void pair(int64_t i, bool in0, bool in1, double (&acc)[2]) const {
        const double al = *alpha;
        const d2 pp = ld2(p, i), aa = ld2(ap, i), xx = ld2(x, i), rr = ld2(r, i);
        const double x0 = xx.a + al * pp.a, x1 = xx.b + al * pp.b;
        const double r0 = rr.a - al * aa.a, r1 = rr.b - al * aa.b;
        st2(x, i, x0, x1); st2(r, i, r0, r1);
        if (in0) { acc[0] = acc[0] + r0 * r0; acc[1] = acc[1] + r0 * pp.a; }
        if (in1) { acc[0] = acc[0] + r1 * r1; acc[1] = acc[1] + r1 * pp.b; }
    }
};
struct CgUpdate0 {                   // no fused dot (PCG with a non-pointwise preconditioner)
    static constexpr int NQ = 0; static constexpr const char* TAG = "CgUpdate0";
    const double* alpha; const double* p; const double* ap; double* x; double* r;
    __device__ __forceinline__ void pair(int64_t i, bool, bool, double (&)[1]) const {
        const double al = *alpha;
        const d2 pp = ld2(p, i), aa = ld2(ap, i), xx = ld2(x, i), rr = ld2(r, i);
        st2(x, i, xx.a + al * pp.a, xx.b + al * pp.b);
        st2(r, i, rr.a - al * aa.a, rr.b - al * aa.b);
    }
};

// ---- the x update deferred to the direction pass.  The reference does x += alpha p (cg.rs:207-209, pcg.rs:175-177) right after alpha;
// nothing reads x again before the solve returns, so here the update rides on the pass that reads p anyway (p = r + beta p, cg.rs:274-276):
// p is read once per iteration instead of twice -- 8 instead of 9 vector passes for CG, 10 instead of 11 for Jacobi-PCG -- and every
// element sees the reference's operations on the reference's operands: x_{k+1} = x_k + alpha_k p_k with p_k not yet overwritten.
// An exit taken after the reference's x update (convergence, iteration cap, indefinite preconditioner) stamps its iteration into
// st->xlast, and that iteration's direction pass still runs its x half (GateXPending); an exit before it (p.Ap <= 0) does not.
// KRYST_CG_DEFER_X=0 keeps the eager form (also used for the Natural norm, the trust region and the objective target, which read p or x
// inside the iteration).
template <bool KEEP = false>
struct CgResidualOp {                // r -= alpha Ap (cg.rs:210-212) ; partial r.r (:223)
    static constexpr int NQ = 1; static constexpr const char* TAG = "CgResidual"; static constexpr int PHASE = KR_PH_BLAS1_RESIDUAL; static constexpr int BPC = 3;       // 2 reads + 1 write + a fold per tile: 3 workgroups per CU (tools/stream_ab.py: 512^3 0.609 -> 0.533 ms, 256^3 0.077 -> 0.065)
    const double* alpha; const double* ap; double* r;
    __device__ __forceinline__ void pair(int64_t i, bool in0, bool in1, double (&acc)[1]) const {
        const double al = *alpha;
        const d2 aa = ld2(ap, i), rr = ld2(r, i);
        const double r0 = rr.a - al * aa.a, r1 = rr.b - al * aa.b;
        st2_sel<KEEP>(r, i, r0, r1);                                      // r is read again by the direction pass
        if (in0) acc[0] = acc[0] + r0 * r0;
        if (in1) acc[0] = acc[0] + r1 * r1;
    }
};
struct CgResidual0Op {               // r -= alpha Ap, no fused dot (PCG with a non-pointwise preconditioner)
    static constexpr int NQ = 0; static constexpr const char* TAG = "CgResidual0";
    const double* alpha; const double* ap; double* r;
    __device__ __forceinline__ void pair(int64_t i, bool, bool, double (&)[1]) const {
        const double al = *alpha;
        const d2 aa = ld2(ap, i), rr = ld2(r, i);
        st2(r, i, rr.a - al * aa.a, rr.b - al * aa.b);
    }
};
template <bool KEEP = false>
struct CgDirectionOp {               // x += alpha p (cg.rs:207-209, deferred) ; p = z + beta p (cg.rs:274-276, pcg.rs:215-217; z = r for CG)
    static constexpr int NQ = 0; static constexpr const char* TAG = "CgDirection"; static constexpr int PHASE = KR_PH_BLAS1_DIRECTION; static constexpr int BPC = 3;   // inside CG: 2 -> 3 +4.4 % (256^3), +1.1 % (512^3); PCG +3.0 % / -0.7 %
    const DevState* st; const double* z; double* p; double* x;
    __device__ __forceinline__ void pair(int64_t i, bool, bool, double (&)[1]) const {
        const double al = st->alpha, be = st->beta;
        const bool ended = st->done != 0;                                 // (uniform) the solve ended in this iteration: only x is still owed
        const d2 pp = ld2(p, i), xx = ld2(x, i);
        st2(x, i, xx.a + al * pp.a, xx.b + al * pp.b);
        if (!ended) {
            const d2 zz = ld2_sel<KEEP>(z, i);
            st2(p, i, zz.a + be * pp.a, zz.b + be * pp.b);
        }
    }
};
// the same pass with the direction vectors in a RING and x updated in batches (XBatchOp below): p_new = z + beta p_old, out of place; x is not touched
template <bool KEEP = false>
struct CgDirectionRingOp {
    static constexpr int NQ = 0; static constexpr const char* TAG = "CgDirectionRing"; static constexpr int PHASE = KR_PH_BLAS1_DIRECTION; static constexpr int BPC = 3;
    const DevState* st; const double* z; const double* p_old; double* p_new;
    __device__ __forceinline__ void pair(int64_t i, bool, bool, double (&)[1]) const {
        if (st->done != 0) return;                                        // (uniform) the solve ended in this iteration: no further direction
        const double be = st->beta;
        const d2 pp = ld2(p_old, i), zz = ld2_sel<KEEP>(z, i);
        st2(p_new, i, zz.a + be * pp.a, zz.b + be * pp.b);
    }
};
struct GateXPending {                // runs while the solve is under way, and once more in the iteration that ended it after the x update
    const DevState* st; long long it;
    __device__ __forceinline__ bool skip() const { return st->done && st->xlast != it; }
};
// The direction pass on a distributed operator whose neighbours get contiguous runs of rows (k-slab partitions): the tiles that hold
// those rows first, then the halo exchange of the new p is STARTED (spmv.hip: halo_begin), then the rest of the vector -- the planes
// travel while the pass is still writing the interior, not only during the next SpMV's interior tiles (which the staged-window
// kernel has made shorter than the exchange).  Same launches' worth of work, same bits.  KRYST_HALO_EARLY=0: one launch, exchange
// started by the SpMV as before.
template <class Op>
inline int32_t launch_direction(kryst_ctx_t ctx, kryst_csr_t a, const Op& op, int64_t n, const DevState* st, long long it, const double* p_vec) {
    const GateXPending gate{st, it};
    // The early start adds one exchange per iteration, so WHETHER it happens must not depend on anything a rank sees alone: it is
    // a->halo_early_ok (agreed across ranks when the operator was created) and the environment (the same on every rank, like every
    // other setting).  Whether this rank SPLITS its pass around the exchange is a local matter: a rank whose send ranges are most of
    // its block runs the whole pass first and starts the exchange behind it -- the same number of exchanges either way.
    if (a->dist && use_collectives(ctx) && a->halo_early_ok && env_int("KRYST_HALO_EARLY", 1) != 0) {
        std::vector<std::pair<int64_t, int64_t>> early;
        halo_send_tiles(a, early);
        int64_t covered = 0;
        for (const auto& r : early) covered += r.second - r.first;
        const int64_t all = ntiles_of(n);
        if (!early.empty() && early.size() <= 4 && 2 * covered < all) {
            for (const auto& r : early) KR_TRY(launch_ew_gated(ctx, op, n, gate, 0, r.first, std::min(r.second, all)));
            KR_TRY(halo_begin(a, p_vec));
            int64_t at = 0;
            for (const auto& r : early) { KR_TRY(launch_ew_gated(ctx, op, n, gate, 0, at, std::min(r.first, all))); at = std::min(r.second, all); }
            return launch_ew_gated(ctx, op, n, gate, 0, at, all);
        }
        KR_TRY(launch_ew_gated(ctx, op, n, gate));
        return halo_begin(a, p_vec);
    }
    return launch_ew_gated(ctx, op, n, gate);
}
inline bool cg_defer_x() { return env_int("KRYST_CG_DEFER_X", 1) != 0; }      // (read per iteration enqueue: a getenv, not on any critical path)

struct CgInitLogic {                 // cg.rs:127-140
    static constexpr bool RUN_WHEN_DONE = false;
    LogicCtx c;
    __device__ void run(const double* red) const {
        DevState* st = c.st;
        st->rsq = red[0];
        st->res0 = dsqrt(st->rsq);
        st->iterations = 0; st->final_residual = st->res0; st->converged = 0; st->iter = 0;
        // dp: Preconditioned/Unpreconditioned = (r,r); Natural = (r,p) with p == r; None = 0
        const double dp = (c.norm_type == 3) ? 0.0 : st->rsq;
        c.push(dsqrt(dp));
        if (c.max_iters <= 0) c.finish(KRYST_OK);
    }
};
struct CgAlphaLogic {                // cg.rs:164-175
    static constexpr bool RUN_WHEN_DONE = false;
    LogicCtx c;
    __device__ void run(const double* red) const {
        DevState* st = c.st;
        const double p_dot_ap = red[0];
        if (p_dot_ap <= 0.0) {                                         // :168-174
            st->iterations = st->iter + 1; st->final_residual = dsqrt(st->rsq); st->converged = 0;
            c.finish(KRYST_INDEFINITE_MATRIX);
            return;
        }
        st->alpha = st->rsq / p_dot_ap;                                // :175
        st->alpha_hist[(st->iter + 1) & 15] = st->alpha;               // (x updated in batches: XBatchOp reads it; the ring of direction vectors has <= 16 slots)
    }
};
struct CgBetaLogic {                 // cg.rs:223-284
    static constexpr bool RUN_WHEN_DONE = false;
    LogicCtx c;
    __device__ void run(const double* red) const {
        DevState* st = c.st;
        const long long i = st->iter + 1;
        st->xpend = i;                                                 // (x += alpha p of this iteration, :207-209, is behind us in the reference)
        const double rsq_new = red[0];
        double res_norm;
        switch (c.norm_type) {                                         // :224-229
            case 0: case 1: res_norm = dsqrt(rsq_new); break;
            case 2: res_norm = dsqrt(fabs(red[1])); break;
            default: res_norm = 0.0;
        }
        if (rsq_new / st->rsq < 0.0) {                                 // :254-259
            st->iterations = i; st->final_residual = res_norm; st->converged = 0;
            st->xlast = i;                                             // (x += alpha p happened at :207)
            c.finish(KRYST_INDEFINITE_PRECONDITIONER);
            return;
        }
        c.push(res_norm);                                              // :260-263
        st->iter = i;
        if (c.check(res_norm, st->res0, i)) { st->xlast = i; c.finish(KRYST_OK); return; }   // :264-269
        st->beta = rsq_new / st->rsq;                                  // :270
        st->rsq = rsq_new;                                             // :284
    }
};

// ---- CG side exits (off by default in the reference): trust region cg.rs:177-202, objective target cg.rs:231-252
struct CgRadiusLogic {               // red0 = (p,p), red1 = (x,x)
    static constexpr bool RUN_WHEN_DONE = false;
    LogicCtx c; double radius;
    __device__ void run(const double* red) const {
        DevState* st = c.st;
        const double p_norm = dsqrt(red[0]), x_norm = dsqrt(red[1]);                       // :178-179
        if (x_norm + fabs(st->alpha) * p_norm > radius) {                                  // :180
            st->alpha = (radius - x_norm) / p_norm;                                        // :181 max_step, applied by AxpyIfOp
            st->iterations = st->iter + 1; st->final_residual = dsqrt(st->rsq); st->converged = 0;   // :196-199
            st->early = 1;
            c.finish(KRYST_OK);
        }
    }
};
struct CgRsqLogic {                  // first half of cg.rs:223-229 when the objective exit sits in between
    static constexpr bool RUN_WHEN_DONE = false;
    LogicCtx c;
    __device__ void run(const double* red) const {
        DevState* st = c.st;
        st->rz = red[0];                                                                    // rsq_new
        switch (c.norm_type) { case 0: case 1: st->normq = dsqrt(red[0]); break; case 2: st->normq = dsqrt(fabs(red[1])); break; default: st->normq = 0.0; }
    }
};
struct StoreLogic {                  // keeps one reduced value for a later logic step
    static constexpr bool RUN_WHEN_DONE = false;
    LogicCtx c; double* slot;
    __device__ void run(const double* red) const { *slot = red[0]; }
};
struct CgObjLogic {                  // cg.rs:231-252 ; red0 = (x,b), *xax = (x,Ax)
    static constexpr bool RUN_WHEN_DONE = false;
    LogicCtx c; const double* xax; double target;
    __device__ void run(const double* red) const {
        DevState* st = c.st;
        const double obj = 0.5 * (*xax) - red[0];                                           // :237
        if (obj <= target) {                                                                // :245-251
            st->iterations = st->iter + 1; st->final_residual = st->normq; st->converged = 1;
            c.finish(KRYST_OK);
        }
    }
};
struct CgBetaStoredLogic {           // second half: cg.rs:254-284 on the stored rsq_new / res_norm
    static constexpr bool RUN_WHEN_DONE = false;
    LogicCtx c;
    __device__ void run(const double*) const {
        DevState* st = c.st;
        const long long i = st->iter + 1;
        const double rsq_new = st->rz, res_norm = st->normq;
        if (rsq_new / st->rsq < 0.0) {
            st->iterations = i; st->final_residual = res_norm; st->converged = 0;
            c.finish(KRYST_INDEFINITE_PRECONDITIONER);
            return;
        }
        c.push(res_norm);
        st->iter = i;
        if (c.check(res_norm, st->res0, i)) { c.finish(KRYST_OK); return; }
        st->beta = rsq_new / st->rsq;
        st->rsq = rsq_new;
    }
};
struct AxpyIfOp {                    // x += alpha*p, only while st->early is raised (cg.rs:185-187 max_step update)
    static constexpr int NQ = 0; static constexpr const char* TAG = "AxpyIf";
    const DevState* st; const double* p; double* x;
    __device__ __forceinline__ void pair(int64_t i, bool, bool, double (&)[1]) const {
        const double al = st->alpha;
        const d2 pp = ld2(p, i), xx = ld2(x, i);
        st2(x, i, xx.a + al * pp.a, xx.b + al * pp.b);
    }
};
// the x update the fused form still owes when no further fused SpMV follows (the end of a solve or session)
struct GateXOwed {
    const DevState* st; long long it;
    __device__ __forceinline__ bool skip() const { return st->xpend != it; }
};
struct CgFlushOp {                   // x += alpha p (cg.rs:207-209 / pcg.rs:175-177) of the last enqueued iteration
    static constexpr int NQ = 0; static constexpr const char* TAG = "CgFlush"; static constexpr int PHASE = KR_PH_BLAS1_DIRECTION;
    const DevState* st; const double* p; double* x;
    __device__ __forceinline__ void pair(int64_t i, bool, bool, double (&)[1]) const {
        const double al = st->alpha;
        const d2 pp = ld2(p, i), xx = ld2(x, i);
        st2(x, i, xx.a + al * pp.a, xx.b + al * pp.b);
    }
};
// ---- x updated in BATCHES (round 5).  A written byte costs this HBM what 2-2.5 read bytes cost (profiles/r05/fuse_kernel_ablations.txt: the fused
// kernel's three written vectors are 0.54 of its 1.34 ms), and x += alpha_i p_i (cg.rs:207-209) writes x every iteration although nothing reads x
// before the solve returns.  With the direction vectors kept in a ring of m + 1 buffers (p_new is written every iteration anyway) the m updates of
// iterations k - m + 1 .. k are applied in ONE pass -- x = x + alpha_i p_i for i ascending, each the reference's un-fused multiply and add on the
// reference's operands, so the same bits -- which reads x once and writes it once per m iterations.  Updates beyond `xpend` (the solve ended before
// them) are not applied; the flush at the end of a solve / session applies what the last partial batch owes.
template <int M>                   // M: the batch's length when every one of its updates happened (the common case), 0: any length, tested per update
struct XBatchOp {
    static constexpr int NQ = 0; static constexpr const char* TAG = "XBatch"; static constexpr int PHASE = KR_PH_BLAS1_XBATCH; static constexpr int BPC = 2;
    const DevState* st; const double* p[8]; long long lo; int cnt; double* x;       // p[u]: the direction vector of iteration lo + u
    __device__ __forceinline__ void pair(int64_t i, bool, bool, double (&)[1]) const {
        const long long last = st->xpend;                                     // (uniform) updates of later iterations never happened in the reference
        d2 xx = ld2(x, i);
        if (M > 0 && lo + M - 1 <= last) {                                    // all M loads in flight together, then the M updates in iteration order
            d2 pp[M > 0 ? M : 1]; double al[M > 0 ? M : 1];
#pragma unroll
            for (int u = 0; u < M; ++u) { pp[u] = ld2(p[u], i); al[u] = st->alpha_hist[(lo + u) & 15]; }
#pragma unroll
            for (int u = 0; u < M; ++u) { xx.a = xx.a + al[u] * pp[u].a; xx.b = xx.b + al[u] * pp[u].b; }
        } else {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (u < cnt && lo + u <= last) {
                    const double al = st->alpha_hist[(lo + u) & 15];
                    const d2 pp = ld2(p[u], i);
                    xx.a = xx.a + al * pp.a; xx.b = xx.b + al * pp.b;
                }
            }
        }
        st2(x, i, xx.a, xx.b);
    }
};
struct GateXBatch {                  // nothing to apply: the solve ended before this batch's first iteration
    const DevState* st; long long lo;
    __device__ __forceinline__ bool skip() const { return st->xpend < lo; }
};
template <int M>
inline int32_t launch_x_batch_m(kryst_ctx_t ctx, DevState* st, const std::vector<double*>& ring, int xb, long long lo, long long hi, double* x, int64_t n) {
    XBatchOp<M> op; op.st = st; op.lo = lo; op.cnt = (int)(hi - lo + 1); op.x = x;
    for (int u = 0; u < 8; ++u) op.p[u] = ring[(size_t)((lo + std::min<long long>(u, hi - lo)) % (xb + 1))];
    return launch_ew_gated(ctx, op, n, GateXBatch{st, lo});
}
// x += alpha_i p_i for iterations lo .. hi (those that happened), p_i in ring[i % (xb + 1)]
inline int32_t launch_x_batch(kryst_ctx_t ctx, DevState* st, const std::vector<double*>& ring, int xb, long long lo, long long hi, double* x, int64_t n) {
    switch ((int)(hi - lo + 1)) {
        case 2: return launch_x_batch_m<2>(ctx, st, ring, xb, lo, hi, x, n);
        case 3: return launch_x_batch_m<3>(ctx, st, ring, xb, lo, hi, x, n);
        case 4: return launch_x_batch_m<4>(ctx, st, ring, xb, lo, hi, x, n);
        case 5: return launch_x_batch_m<5>(ctx, st, ring, xb, lo, hi, x, n);
        case 6: return launch_x_batch_m<6>(ctx, st, ring, xb, lo, hi, x, n);
        case 7: return launch_x_batch_m<7>(ctx, st, ring, xb, lo, hi, x, n);
        case 8: return launch_x_batch_m<8>(ctx, st, ring, xb, lo, hi, x, n);
        default: return launch_x_batch_m<0>(ctx, st, ring, xb, lo, hi, x, n);
    }
}
// x-batch length: 1 = every iteration (x rides on the fused pass that reads p_old); KRYST_CG_X_BATCH forces (<= 8).  Measured at 512^3
// (profiles/r05/cg_xbatch_ab.jsonl): CG 545 / 560 / 575 / 579 / 584 / 582 / 588 it/s at m = 1 / 2 / 4 / 5 / 6 / 7 / 8 (unfused 530); PCG 447 / 459 / 487 /
// 489 / 490 / 490.5 / 472 (unfused 443) -- PCG's thirteen 1 GB work vectors lose at m = 8 what the ninth ring slot costs, so 7 there.
inline int cg_x_batch(bool pcg) {
    return std::max(1, std::min(8, env_int("KRYST_CG_X_BATCH", pcg ? 7 : 8)));
}
// the ring costs xb - 1 work vectors more than the two alternating direction vectors: only where the device has the room (what the context's
// arena already holds counts as room)
inline int ring_if_it_fits(kryst_ctx_t ctx, int64_t n, int xb, int work_vectors) {
    if (xb <= 1) return 1;
    size_t free_b = 0, total_b = 0;
    if (hipSetDevice(ctx->device) != hipSuccess || hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void)hipGetLastError(); return 1; }
    const size_t need = padded_bytes(n) * (size_t)(work_vectors + 2);
    return need <= ctx->arena_bytes || need <= free_b + ctx->arena_bytes ? xb : 1;
}
// ... and in the UNFUSED deferred form (every operator form, every rank of a partition) the same ring is available: the direction pass writes
// p_new = z + beta p_old out of place (CgDirectionRingOp: 2 reads + 1 write instead of 3 + 2) and XBatchOp pays x -- 34 instead of 40 bytes per row
// and iteration.  MEASURED AND OFF BY DEFAULT (profiles/r05/cg_ring_unfused_ab.jsonl): 192^3 .. 384^3 run 2-7 % SLOWER (the in-place direction vector
// is what the 256 MiB Infinity Cache hands from the direction pass to the SpMV; nine ring slots defeat that), 512^3 on plain CSR +-1 %, 64^3 / 128^3
// +3 %.  KRYST_CG_X_BATCH = m > 1 turns it on (tests do).
inline int cg_x_batch_unfused() {
    return std::max(1, std::min(8, env_int("KRYST_CG_X_BATCH", 1)));
}
struct CgRun : SolverRun {
    using SolverRun::SolverRun;
    double *r = nullptr, *pp = nullptr, *ap = nullptr, *ax = nullptr;
    double* p2 = nullptr;            // the fused form's second direction vector (p_old / p_new alternate)
    bool fuse = false; long long fused_upto = 0;     // fused_upto: the last iteration enqueued in the fused form (its x update rides on the next one)
    int xb = 1; std::vector<double*> ring;           // xb > 1: x in batches of xb iterations, direction vector of iteration k in ring[k % (xb + 1)]
    bool ringdir = false;                            // the unfused deferred form with the ring (CgDirectionRingOp)
    int32_t x_batch(long long lo, long long hi) {    // apply the x updates of iterations lo .. hi (those that happened)
        if (hi < lo) return KRYST_OK;
        return launch_x_batch(ctx, ws.st, ring, xb, lo, hi, xw, n);
    }
    int32_t begin() override {
        KR_TRY(solve_args_check(io, bv, xv));
        // the direction pass inside the SpMV (spmv.hip: spmv_pattern_fuse_kernel) where the operator's form allows it
        const bool deferred = cg_defer_x() && !prm.has_radius && !prm.has_obj_target && prm.norm_type != 2;
        fuse = deferred && spmv_can_fuse_direction(a);
        xb = fuse ? cg_x_batch(false) : deferred ? cg_x_batch_unfused() : 1;
        xb = ring_if_it_fits(ctx, n, xb, 3 + xb + 1);
        ringdir = !fuse && xb > 1;
        KR_TRY(common_begin(prm.max_iters + 2, xb > 1 ? 3 + xb + 1 : fuse ? 5 : 4));              // cg.rs:117
        KR_TRY(ws.vec(&r)); KR_TRY(ws.vec(&pp)); KR_TRY(ws.vec(&ap));
        if (xb > 1) {                                  // the ring: slot 1 is p_1 (= pp below), the others follow
            ring.assign((size_t)xb + 1, nullptr);
            ring[1 % (xb + 1)] = pp;
            for (int k = 0; k <= xb; ++k) if (!ring[(size_t)k]) KR_TRY(ws.vec(&ring[(size_t)k]));
        } else if (fuse) KR_TRY(ws.vec(&p2));
        if (prm.has_obj_target) KR_TRY(ws.vec(&ax));
        KR_TRY(residual_dot(a, bv->d, xw, r, ap, nullptr));                                       // :120-125, :127
        KR_HIP(hipMemcpyAsync(pp, r, padded_bytes(n), hipMemcpyDeviceToDevice, ctx->s_main));      // :126
        return reduce_then<1>(ctx, nt, ws.red, CgInitLogic{lc});
    }
    int32_t flush() override {
        if ((!fuse && !ringdir) || fused_upto == 0) return KRYST_OK;
        if (xb > 1) return x_batch(fused_upto / xb * xb + 1, fused_upto);                         // what the last partial batch owes
        return launch_ew_gated(ctx, CgFlushOp{ws.st, pp, xw}, n, GateXOwed{ws.st, fused_upto});
    }
    int32_t iterate(int64_t it) override {
        if (fuse) {
            // iteration it: [x += alpha p_old owed by iteration it - 1 (xb == 1); p = r + beta p_old; Ap; (p, Ap)] in ONE pass, alpha, the residual pass, beta
            if (fused_upto == it - 1 && it > 1) {
                if (xb > 1) {
                    double* p_old = ring[(size_t)((it - 1) % (xb + 1))]; double* p_new = ring[(size_t)(it % (xb + 1))];
                    KR_TRY(launch_spmv_fused(a, r, p_old, p_new, nullptr, ap, 1, &ws.st->alpha, &ws.st->beta, &ws.st->xpend, (long long)it, done));
                    pp = p_new;
                } else {
                    KR_TRY(launch_spmv_fused(a, r, pp, p2, xw, ap, 1, &ws.st->alpha, &ws.st->beta, &ws.st->xpend, (long long)it, done));
                    std::swap(pp, p2);
                }
            } else {
                KR_TRY(launch_spmv(a, pp, ap, 1, pp, done));                                      // the first iteration: p = r already
            }
            KR_TRY((reduce_then<1>(ctx, nt, ws.red, CgAlphaLogic{lc})));
            if (keep_in_cache(n)) KR_TRY(launch_ew(ctx, CgResidualOp<true>{&ws.st->alpha, ap, r}, n, done));
            else KR_TRY(launch_ew(ctx, CgResidualOp<false>{&ws.st->alpha, ap, r}, n, done));
            KR_TRY((reduce_then<1>(ctx, nt, ws.red, CgBetaLogic{lc})));
            fused_upto = it;
            if (xb > 1 && it % xb == 0) KR_TRY(x_batch(it - xb + 1, it));                          // the x updates of the last xb iterations, one pass
            return KRYST_OK;
        }
        KR_TRY(launch_spmv(a, pp, ap, 1, pp, done));                                              // :143-144 + (p,Ap) :164
        KR_TRY((reduce_then<1>(ctx, nt, ws.red, CgAlphaLogic{lc})));
        if (cg_defer_x() && !prm.has_radius && !prm.has_obj_target && prm.norm_type != 2) {       // x += alpha p rides on the direction pass
            if (keep_in_cache(n)) KR_TRY(launch_ew(ctx, CgResidualOp<true>{&ws.st->alpha, ap, r}, n, done));
            else KR_TRY(launch_ew(ctx, CgResidualOp<false>{&ws.st->alpha, ap, r}, n, done));      // :210-212 + (r,r) :223
            KR_TRY((reduce_then<1>(ctx, nt, ws.red, CgBetaLogic{lc})));
            if (ringdir) {                             // p_{it+1} = r + beta p_it into the next ring slot; x in batches
                double* p_new = ring[(size_t)((it + 1) % (xb + 1))];
                if (keep_in_cache(n)) KR_TRY(launch_direction(ctx, a, CgDirectionRingOp<true>{ws.st, r, pp, p_new}, n, ws.st, (long long)it, p_new));
                else KR_TRY(launch_direction(ctx, a, CgDirectionRingOp<false>{ws.st, r, pp, p_new}, n, ws.st, (long long)it, p_new));
                pp = p_new; fused_upto = it;
                if (it % xb == 0) KR_TRY(x_batch(it - xb + 1, it));
                return KRYST_OK;
            }
            if (keep_in_cache(n)) return launch_direction(ctx, a, CgDirectionOp<true>{ws.st, r, pp, xw}, n, ws.st, (long long)it, pp);
            return launch_direction(ctx, a, CgDirectionOp<false>{ws.st, r, pp, xw}, n, ws.st, (long long)it, pp);   // :207-209, :274-276
        }
        if (prm.has_radius) {                                                                     // :177-202 (Steihaug-Toint)
            KR_TRY(launch_ew(ctx, DotPairOp{pp, pp, xw, xw}, n, done));
            KR_TRY((reduce_then<2>(ctx, nt, ws.red, CgRadiusLogic{lc, prm.radius})));
            KR_TRY(launch_ew_gated(ctx, AxpyIfOp{ws.st, pp, xw}, n, GateIfEarly{ws.st}));
            hipLaunchKernelGGL((logic_kernel<ClearEarlyLogic>), dim3(1), dim3(64), 0, ctx->s_main, ws.red, ClearEarlyLogic{lc});
            KR_HIP(hipGetLastError());
        }
        const bool nat = prm.norm_type == 2;
        if (nat) KR_TRY(launch_ew(ctx, CgUpdate2{&ws.st->alpha, pp, ap, xw, r}, n, done));
        else if (keep_in_cache(n)) KR_TRY(launch_ew(ctx, CgUpdate1<true>{&ws.st->alpha, pp, ap, xw, r}, n, done));
        else KR_TRY(launch_ew(ctx, CgUpdate1<false>{&ws.st->alpha, pp, ap, xw, r}, n, done));            // :207-212 + (r,r) :223
        if (!prm.has_obj_target) {
            if (nat) KR_TRY((reduce_then<2>(ctx, nt, ws.red, CgBetaLogic{lc})));
            else KR_TRY((reduce_then<1>(ctx, nt, ws.red, CgBetaLogic{lc})));
        } else {                                                                                  // :231-252
            if (nat) KR_TRY((reduce_then<2>(ctx, nt, ws.red, CgRsqLogic{lc})));
            else KR_TRY((reduce_then<1>(ctx, nt, ws.red, CgRsqLogic{lc})));
            KR_TRY(launch_spmv(a, xw, ax, 1, xw, done));                                          // ax = A x, (x, Ax)
            KR_TRY((reduce_then<1>(ctx, nt, ws.red, StoreLogic{lc, &ws.st->omega})));
            KR_TRY(launch_ew(ctx, DotOneOp{xw, bv->d}, n, done));                                 // (x, b)
            KR_TRY((reduce_then<1>(ctx, nt, ws.red, CgObjLogic{lc, &ws.st->omega, prm.obj_target})));
            hipLaunchKernelGGL((logic_kernel<CgBetaStoredLogic>), dim3(1), dim3(64), 0, ctx->s_main, ws.red, CgBetaStoredLogic{lc});
            KR_HIP(hipGetLastError());
        }
        if (keep_in_cache(n)) return launch_ew(ctx, AypxDevOp<true>{&ws.st->beta, r, pp}, n, done);
        return launch_ew(ctx, AypxDevOp<false>{&ws.st->beta, r, pp}, n, done);                           // :274-276
    }
};

int32_t cg_solve(kryst_vec_t bv, kryst_vec_t xv, const SolveIO& io) {
    KR_ARG(io.a && io.params, "solve: null argument");
    CgRun run(bv, xv, io);
    return run.solve();
}

// =================================================================== PCG (src/solver/pcg.rs:114-222)
// fused update for z == r (pc None / identity) and z = D^-1 r (Jacobi):
//   x += alpha p ; r -= alpha Ap ; z = M^-1 r ; partial 0 = r.z ; partial 1 = norm quantity (z.z | r.r)
template <bool JACOBI, bool KEEP = false>
struct PcgUpdateOp {
    static constexpr int NQ = 2; static constexpr const char* TAG = "PcgUpdate"; static constexpr int PHASE = KR_PH_BLAS1_RESIDUAL; static constexpr int BPC = 3;       // 5 reads + 3 writes: 3 workgroups per CU measured best (+3 %)
    const double* alpha; const double* p; const double* ap; double* x; double* r; double* z; const double* inv;
    int norm_type;
    __device__ __forceinline__ void pair(int64_t i, bool in0, bool in1, double (&acc)[2]) const {
        const double al = *alpha;
        const d2 pp = ld2(p, i), aa = ld2(ap, i), xx = ld2(x, i), rr = ld2(r, i);
        const double x0 = xx.a + al * pp.a, x1 = xx.b + al * pp.b;                   // pcg.rs:175-177
        const double r0 = rr.a - al * aa.a, r1 = rr.b - al * aa.b;                   // pcg.rs:179-181
        st2(x, i, x0, x1); st2(r, i, r0, r1);
        double z0 = r0, z1 = r1;
        if constexpr (JACOBI) {
            const d2 dv = ld2(inv, i);
            z0 = dv.a * r0; z1 = dv.b * r1;                                          // jacobi.rs:84-86
            st2_sel<KEEP>(z, i, z0, z1);                                             // z is read again by the next launch (p = z + beta p)
        }
        const bool zz = norm_type == 0;                                              // Preconditioned: (z,z); else (r,r)
        if (in0) { acc[0] = acc[0] + r0 * z0; acc[1] = acc[1] + (zz ? z0 * z0 : r0 * r0); }
        if (in1) { acc[0] = acc[0] + r1 * z1; acc[1] = acc[1] + (zz ? z1 * z1 : r1 * r1); }
    }
};

template <bool JACOBI, bool KEEP = false>
struct PcgResidualOp {               // PcgUpdateOp without its x half (deferred to CgDirectionOp): 2-3 reads + 1-2 writes
    static constexpr int NQ = 2; static constexpr const char* TAG = "PcgResidual"; static constexpr int PHASE = KR_PH_BLAS1_RESIDUAL; static constexpr int BPC = 3;
    const double* alpha; const double* ap; double* r; double* z; const double* inv;
    int norm_type;
    __device__ __forceinline__ void pair(int64_t i, bool in0, bool in1, double (&acc)[2]) const {
        const double al = *alpha;
        const d2 aa = ld2(ap, i), rr = ld2(r, i);
        const double r0 = rr.a - al * aa.a, r1 = rr.b - al * aa.b;                   // pcg.rs:179-181
        double z0 = r0, z1 = r1;
        if constexpr (JACOBI) {
            st2(r, i, r0, r1);
            const d2 dv = ld2(inv, i);
            z0 = dv.a * r0; z1 = dv.b * r1;                                          // jacobi.rs:84-86
            st2_sel<KEEP>(z, i, z0, z1);                                             // z is read again by the direction pass
        } else {
            st2_sel<KEEP>(r, i, r0, r1);                                             // z == r
        }
        const bool zz = norm_type == 0;                                              // Preconditioned: (z,z); else (r,r)
        if (in0) { acc[0] = acc[0] + r0 * z0; acc[1] = acc[1] + (zz ? z0 * z0 : r0 * r0); }
        if (in1) { acc[0] = acc[0] + r1 * z1; acc[1] = acc[1] + (zz ? z1 * z1 : r1 * r1); }
    }
};

struct PcgInitLogic {                // pcg.rs:133-146 ; red0 = (r,z), red1 = (z,z) | (r,r)
    static constexpr bool RUN_WHEN_DONE = false;
    LogicCtx c;
    __device__ void run(const double* red) const {
        DevState* st = c.st;
        st->rz = red[0];
        st->res0 = dsqrt(fabs(st->rz));                                // :134
        st->iterations = 0; st->final_residual = st->res0; st->converged = 0; st->iter = 0;
        double dp;
        switch (c.norm_type) { case 0: case 1: dp = red[1]; break; case 2: dp = red[0]; break; default: dp = 0.0; }
        st->normq = dp;
        c.push(dsqrt(dp));                                             // :143-146 (no abs at iteration 0)
        if (c.max_iters <= 0) c.finish(KRYST_OK);
    }
};
struct PcgAlphaLogic {               // pcg.rs:151-173
    static constexpr bool RUN_WHEN_DONE = false;
    LogicCtx c;
    __device__ void run(const double* red) const {
        DevState* st = c.st;
        const double p_dot_ap = red[0];
        if (p_dot_ap <= 0.0) {                                         // :162-172 (the dots of the unchanged r, z)
            st->iterations = st->iter + 1;
            st->final_residual = (c.norm_type == 3) ? 0.0 : dsqrt(c.norm_type == 2 ? fabs(st->normq) : st->normq);
            st->converged = 0;
            c.finish(KRYST_INDEFINITE_MATRIX);
            return;
        }
        st->alpha = st->rz / p_dot_ap;                                 // :173
        st->alpha_hist[(st->iter + 1) & 15] = st->alpha;
    }
};
struct PcgBetaLogic {                // pcg.rs:188-218
    static constexpr bool RUN_WHEN_DONE = false;
    LogicCtx c;
    __device__ void run(const double* red) const {
        DevState* st = c.st;
        const long long i1 = st->iter + 1;                             // the reference's i + 1
        st->xpend = i1;                                                // (x += alpha p of this iteration, :175-177, is behind us in the reference)
        const double rz_new = red[0];
        double res_norm;
        switch (c.norm_type) {                                         // :190-195
            case 0: case 1: res_norm = dsqrt(red[1]); st->normq = red[1]; break;
            case 2: res_norm = dsqrt(fabs(rz_new)); st->normq = rz_new; break;
            default: res_norm = 0.0;
        }
        c.push(res_norm);                                              // :196-199
        st->iter = i1;
        if (c.check(res_norm, st->res0, i1)) { st->xlast = i1; c.finish(KRYST_OK); return; }   // :200-205 (x += alpha p happened at :175)
        const double beta = rz_new / st->rz;                           // :206
        if (beta < 0.0) {                                              // :208-213
            st->iterations = i1; st->final_residual = res_norm; st->converged = 0;
            st->xlast = i1;
            c.finish(KRYST_INDEFINITE_PRECONDITIONER);
            return;
        }
        st->beta = beta;
        st->rz = rz_new;                                               // :218
    }
};

struct PcgRun : SolverRun {
    using SolverRun::SolverRun;
    double *r = nullptr, *z = nullptr, *pp = nullptr, *ap = nullptr;
    double* p2 = nullptr;            // the fused form's second direction vector
    bool alias = false, jac = false;
    bool fuse = false; long long fused_upto = 0;
    int xb = 1; std::vector<double*> ring;           // x in batches (see CgRun)
    bool ringdir = false;
    int32_t x_batch(long long lo, long long hi) {
        if (hi < lo) return KRYST_OK;
        return launch_x_batch(ctx, ws.st, ring, xb, lo, hi, xw, n);
    }
    int32_t begin() override {
        KR_TRY(solve_args_check(io, bv, xv));
        // radius / obj_target are fields of PcgSolver (pcg.rs:39-41) but PcgSolver::solve never reads them: accepted, ignored
        fuse = cg_defer_x() && spmv_can_fuse_direction(a);                                        // (spmv.hip: spmv_pattern_fuse_kernel)
        xb = fuse ? cg_x_batch(true) : cg_defer_x() ? cg_x_batch_unfused() : 1;
        xb = ring_if_it_fits(ctx, n, xb, 4 + xb + 1);
        ringdir = !fuse && xb > 1;
        KR_TRY(common_begin(prm.max_iters + 2, xb > 1 ? 4 + xb + 1 : fuse ? 5 : 4));              // pcg.rs:117
        alias = !pc || pc->kind == KR_PC_IDENTITY;      // z == r  (pcg.rs:130,186 clone_from / IdentityPC)
        jac = pc && pc->kind == KR_PC_JACOBI;
        KR_TRY(ws.vec(&r)); KR_TRY(ws.vec(&pp)); KR_TRY(ws.vec(&ap));
        if (xb > 1) {
            ring.assign((size_t)xb + 1, nullptr);
            ring[1 % (xb + 1)] = pp;
            for (int k = 0; k <= xb; ++k) if (!ring[(size_t)k]) KR_TRY(ws.vec(&ring[(size_t)k]));
        } else if (fuse) KR_TRY(ws.vec(&p2));
        if (alias) z = r; else KR_TRY(ws.vec(&z));
        KR_TRY(launch_spmv(a, xw, ap, 0, nullptr, nullptr));                                      // :119-124
        KR_TRY(launch_ew(ctx, SubDotOp{bv->d, ap, r}, n, nullptr));
        if (!alias) KR_TRY(pc_apply_dev(pc, r, z, nullptr));                                      // :127-131 (`?`)
        KR_HIP(hipMemcpyAsync(pp, z, padded_bytes(n), hipMemcpyDeviceToDevice, ctx->s_main));      // :132
        const double* nq_a = (prm.norm_type == 0) ? z : r;      // Preconditioned: (z,z); Unpreconditioned: (r,r)
        KR_TRY(launch_ew(ctx, DotPairOp{r, z, nq_a, nq_a}, n, nullptr));
        return reduce_then<2>(ctx, nt, ws.red, PcgInitLogic{lc});
    }
    int32_t flush() override {
        if ((!fuse && !ringdir) || fused_upto == 0) return KRYST_OK;
        if (xb > 1) return x_batch(fused_upto / xb * xb + 1, fused_upto);
        return launch_ew_gated(ctx, CgFlushOp{ws.st, pp, xw}, n, GateXOwed{ws.st, fused_upto});
    }
    int32_t iterate(int64_t it) override {
        const int nt_ = prm.norm_type;
        const bool fused_now = fuse && fused_upto == it - 1 && it > 1;
        if (fused_now && xb > 1) {
            double* p_old = ring[(size_t)((it - 1) % (xb + 1))]; double* p_new = ring[(size_t)(it % (xb + 1))];
            KR_TRY(launch_spmv_fused(a, z, p_old, p_new, nullptr, ap, 1, &ws.st->alpha, &ws.st->beta, &ws.st->xpend, (long long)it, done));
            pp = p_new;
        } else if (fused_now) {       // [x += alpha p_old owed by iteration it - 1; p = z + beta p_old; Ap; (p, Ap)] in one pass
            KR_TRY(launch_spmv_fused(a, z, pp, p2, xw, ap, 1, &ws.st->alpha, &ws.st->beta, &ws.st->xpend, (long long)it, done));
            std::swap(pp, p2);
        } else {
            KR_TRY(launch_spmv(a, pp, ap, 1, pp, done));                                          // :149-160
        }
        KR_TRY((reduce_then<1>(ctx, nt, ws.red, PcgAlphaLogic{lc})));
        if (cg_defer_x()) {                                                                       // x += alpha p (:175-177) rides on the direction pass
            const bool keep = keep_in_cache(n);
            if (alias) {
                if (keep) KR_TRY(launch_ew(ctx, PcgResidualOp<false, true>{&ws.st->alpha, ap, r, z, nullptr, nt_}, n, done));
                else KR_TRY(launch_ew(ctx, PcgResidualOp<false, false>{&ws.st->alpha, ap, r, z, nullptr, nt_}, n, done));
            } else if (jac) {
                if (keep) KR_TRY(launch_ew(ctx, PcgResidualOp<true, true>{&ws.st->alpha, ap, r, z, pc->d_inv_diag, nt_}, n, done));
                else KR_TRY(launch_ew(ctx, PcgResidualOp<true, false>{&ws.st->alpha, ap, r, z, pc->d_inv_diag, nt_}, n, done));
            } else {
                KR_TRY(launch_ew(ctx, CgResidual0Op{&ws.st->alpha, ap, r}, n, done));             // :179-181
                KR_TRY(pc_apply_dev(pc, r, z, done));                                             // :183-187
                const double* nq_a = (nt_ == 0) ? z : r;
                KR_TRY(launch_ew(ctx, DotPairOp{r, z, nq_a, nq_a}, n, done));                     // :188-195
            }
            KR_TRY((reduce_then<2>(ctx, nt, ws.red, PcgBetaLogic{lc})));
            if (fuse) {                                                                           // (the direction pass is the next iteration's SpMV)
                fused_upto = it;
                if (xb > 1 && it % xb == 0) KR_TRY(x_batch(it - xb + 1, it));
                return KRYST_OK;
            }
            if (ringdir) {
                double* p_new = ring[(size_t)((it + 1) % (xb + 1))];
                if ((alias || jac) && keep) KR_TRY(launch_direction(ctx, a, CgDirectionRingOp<true>{ws.st, z, pp, p_new}, n, ws.st, (long long)it, p_new));
                else KR_TRY(launch_direction(ctx, a, CgDirectionRingOp<false>{ws.st, z, pp, p_new}, n, ws.st, (long long)it, p_new));
                pp = p_new; fused_upto = it;
                if (it % xb == 0) KR_TRY(x_batch(it - xb + 1, it));
                return KRYST_OK;
            }
            if ((alias || jac) && keep) return launch_direction(ctx, a, CgDirectionOp<true>{ws.st, z, pp, xw}, n, ws.st, (long long)it, pp);
            return launch_direction(ctx, a, CgDirectionOp<false>{ws.st, z, pp, xw}, n, ws.st, (long long)it, pp);   // :175-177, :215-217
        }
        if (alias) {
            KR_TRY(launch_ew(ctx, PcgUpdateOp<false>{&ws.st->alpha, pp, ap, xw, r, z, nullptr, nt_}, n, done));
        } else if (jac) {
            if (keep_in_cache(n)) KR_TRY(launch_ew(ctx, PcgUpdateOp<true, true>{&ws.st->alpha, pp, ap, xw, r, z, pc->d_inv_diag, nt_}, n, done));
            else KR_TRY(launch_ew(ctx, PcgUpdateOp<true, false>{&ws.st->alpha, pp, ap, xw, r, z, pc->d_inv_diag, nt_}, n, done));
        } else {
            KR_TRY(launch_ew(ctx, CgUpdate0{&ws.st->alpha, pp, ap, xw, r}, n, done));             // :175-181
            KR_TRY(pc_apply_dev(pc, r, z, done));                                                 // :183-187
            const double* nq_a = (nt_ == 0) ? z : r;
            KR_TRY(launch_ew(ctx, DotPairOp{r, z, nq_a, nq_a}, n, done));                         // :188-195
        }
        KR_TRY((reduce_then<2>(ctx, nt, ws.red, PcgBetaLogic{lc})));
        if (jac && !alias && keep_in_cache(n)) return launch_ew(ctx, AypxDevOp<true>{&ws.st->beta, z, pp}, n, done);
        return launch_ew(ctx, AypxDevOp<false>{&ws.st->beta, z, pp}, n, done);                           // :215-217
    }
};

int32_t pcg_solve(kryst_vec_t bv, kryst_vec_t xv, const SolveIO& io) {
    KR_ARG(io.a && io.params, "solve: null argument");
    PcgRun run(bv, xv, io);
    return run.solve();
}

// =================================================================== BiCGStab (src/solver/bicgstab.rs:69-293)
template <bool KEEP = false>
struct BicgPOp {                     // p = r + beta*(p - omega_prev*v)   (bicgstab.rs:134/140)
    static constexpr int NQ = 0; static constexpr const char* TAG = "BicgP";
    const DevState* st; const double* r; const double* v; double* p;
    __device__ __forceinline__ void pair(int64_t i, bool, bool, double (&)[1]) const {
        const double be = st->beta, om = st->omega_prev;
        const d2 rr = ld2_sel<KEEP>(r, i), vv = ld2(v, i), pp = ld2(p, i);      // r was written by the launch before this one
        st2(p, i, rr.a + be * (pp.a - om * vv.a), rr.b + be * (pp.b - om * vv.b));
    }
};
struct BicgSOp {                     // s = r - alpha*v ; partial s.s     (bicgstab.rs:166-188)
    static constexpr int NQ = 1; static constexpr const char* TAG = "BicgS"; static constexpr int BPC = 4;      // tools/solver_ab.py: 2 -> 4 workgroups per CU, BiCGStab +4.5 % (256^3), +2.0 % (512^3)
    const DevState* st; const double* r; const double* v; double* s;
    __device__ __forceinline__ void pair(int64_t i, bool in0, bool in1, double (&acc)[1]) const {
        const double al = st->alpha;
        const d2 rr = ld2(r, i), vv = ld2(v, i);
        const double s0 = rr.a - al * vv.a, s1 = rr.b - al * vv.b;
        st2(s, i, s0, s1);
        if (in0) acc[0] = acc[0] + s0 * s0;
        if (in1) acc[0] = acc[0] + s1 * s1;
    }
};
// x = x + alpha*p + omega*s ; r = s - omega*t ; partials r.r and rhat.r (the next rho)   (bicgstab.rs:240-279,105-116)
// when the s-norm exit is pending (st->early): only x = x + alpha*p   (bicgstab.rs:191-202)
template <bool KEEP = false>
struct BicgXROp {
    static constexpr int NQ = 2; static constexpr const char* TAG = "BicgXR"; static constexpr int BPC = 3;     // 2 -> 3: +5.2 % (256^3), +3.0 % (512^3)
    // p / sx: the directions x is updated with (M^-1 p, M^-1 s in the right-preconditioned extension); s: the true s
    const DevState* st; const double* p; const double* sx; const double* s; const double* t; const double* rhat; double* x; double* r;
    __device__ __forceinline__ void pair(int64_t i, bool in0, bool in1, double (&acc)[2]) const {
        const double al = st->alpha;
        const d2 pp = ld2(p, i), xx = ld2(x, i);
        if (st->early) { st2(x, i, xx.a + al * pp.a, xx.b + al * pp.b); return; }
        const double om = st->omega;
        const d2 ss = ld2(s, i), tt = ld2(t, i), hh = ld2(rhat, i);
        d2 sd = ss;
        if (sx != s) sd = ld2(sx, i);
        st2(x, i, xx.a + al * pp.a + om * sd.a, xx.b + al * pp.b + om * sd.b);
        const double r0 = ss.a - om * tt.a, r1 = ss.b - om * tt.b;
        st2_sel<KEEP>(r, i, r0, r1);
        if (in0) { acc[0] = acc[0] + r0 * r0; acc[1] = acc[1] + hh.a * r0; }
        if (in1) { acc[0] = acc[0] + r1 * r1; acc[1] = acc[1] + hh.b * r1; }
    }
};
struct BicgInitLogic {               // bicgstab.rs:79-102 ; red0 = (r,r) = (rhat,r)
    static constexpr bool RUN_WHEN_DONE = false;
    LogicCtx c;
    __device__ void run(const double* red) const {
        DevState* st = c.st;
        st->res0 = dsqrt(red[0]);
        st->iterations = 0; st->final_residual = st->res0; st->converged = 0; st->iter = 0;
        st->rho_prev = 1.0; st->alpha = 1.0; st->omega_prev = 1.0;
        c.push(st->res0);
        if (st->res0 <= c.tol) { st->converged = 1; c.finish(KRYST_OK); return; }   // :98-102 absolute tolerance
        if (c.max_iters <= 0) { c.finish(KRYST_OK); return; }
        st->rho = red[0];                                              // i = 1: rho = (rhat, r), rhat == r
        if (fabs(st->rho) < DBL_EPSILON) { c.finish(KRYST_OK); return; }   // :117-119 break
        st->beta = 0.0;                                                // :120-121
    }
};
struct BicgAlphaLogic {              // bicgstab.rs:149-164 ; red0 = (rhat, v)
    static constexpr bool RUN_WHEN_DONE = false;
    LogicCtx c;
    __device__ void run(const double* red) const {
        DevState* st = c.st;
        const double alpha_den = red[0];
        if (fabs(alpha_den) < DBL_EPSILON) { c.finish(KRYST_OK); return; }   // :161-163 break (stats unchanged)
        st->alpha = st->rho / alpha_den;
    }
};
struct BicgSLogic {                  // bicgstab.rs:177-206 ; red0 = (s,s)
    static constexpr bool RUN_WHEN_DONE = false;
    LogicCtx c;
    __device__ void run(const double* red) const {
        DevState* st = c.st;
        const double s_norm = dsqrt(red[0]);
        if (s_norm <= c.tol) {
            st->iterations = st->iter + 1; st->final_residual = s_norm; st->converged = 1;
            c.push(s_norm);
            st->early = 1;
            c.finish(KRYST_OK);
        }
    }
};
struct BicgOmegaLogic {              // bicgstab.rs:211-238 ; red0 = (t,s), red1 = (t,t)
    static constexpr bool RUN_WHEN_DONE = false;
    LogicCtx c;
    __device__ void run(const double* red) const {
        DevState* st = c.st;
        if (fabs(red[1]) < DBL_EPSILON) { c.finish(KRYST_OK); return; }   // :235-237 break
        st->omega = red[0] / red[1];
    }
};
struct BicgEndLogic {                // bicgstab.rs:268-289 then the head of the next iteration :105-124
    static constexpr bool RUN_WHEN_DONE = true;
    LogicCtx c;
    __device__ void run(const double* red) const {
        DevState* st = c.st;
        if (st->early) { st->early = 0; return; }
        if (st->done) return;
        const long long i = st->iter + 1;
        const double r_norm = dsqrt(red[0]);
        st->iterations = i; st->final_residual = r_norm; st->converged = (r_norm <= c.tol) ? 1 : 0;   // :280
        c.push(r_norm);
        st->iter = i;
        if (r_norm <= c.tol) { c.finish(KRYST_OK); return; }           // :281-284
        if (fabs(st->omega) < DBL_EPSILON) { c.finish(KRYST_OK); return; }   // :285-287 break
        st->rho_prev = st->rho; st->omega_prev = st->omega;           // :288-289
        if (i >= c.max_iters) { c.finish(KRYST_OK); return; }          // loop `1..=max_iters` exhausted
        st->rho = red[1];                                              // :105-116
        if (fabs(st->rho) < DBL_EPSILON) { c.finish(KRYST_OK); return; }   // :117-119
        st->beta = (st->rho / st->rho_prev) * (st->alpha / st->omega_prev);   // :123
    }
};

struct BicgRun : SolverRun {
    bool right_pc;
    double *r = nullptr, *rhat = nullptr, *v = nullptr, *pp = nullptr, *s = nullptr, *t = nullptr, *ph = nullptr, *sh = nullptr;
    BicgRun(kryst_vec_t b, kryst_vec_t x, const SolveIO& io_, bool rp) : SolverRun(b, x, io_), right_pc(rp) {}
    int32_t begin() override {
        KR_TRY(solve_args_check(io, bv, xv));
        KR_TRY(common_begin(prm.max_iters + 2, 8));                                               // :73
        if (!right_pc) pc = nullptr;                     // bicgstab.rs:70: the reference ignores pc; the _rpc extension uses it
        if (pc && pc->kind == KR_PC_IDENTITY) pc = nullptr;
        KR_TRY(ws.vec(&r)); KR_TRY(ws.vec(&rhat)); KR_TRY(ws.vec(&v)); KR_TRY(ws.vec(&pp)); KR_TRY(ws.vec(&s)); KR_TRY(ws.vec(&t));
        if (pc) { KR_TRY(ws.vec(&ph)); KR_TRY(ws.vec(&sh)); }
        KR_TRY(residual_dot(a, bv->d, xw, r, t, nullptr));                                        // :75-77, :85-96
        KR_HIP(hipMemcpyAsync(rhat, r, padded_bytes(n), hipMemcpyDeviceToDevice, ctx->s_main));    // :78
        KR_HIP(hipMemcpyAsync(pp, r, padded_bytes(n), hipMemcpyDeviceToDevice, ctx->s_main));      // :83
        return reduce_then<1>(ctx, nt, ws.red, BicgInitLogic{lc});
    }
    int32_t iterate(int64_t) override {
        const DevState* st = ws.st;
        if (keep_in_cache(n)) KR_TRY(launch_ew(ctx, BicgPOp<true>{st, r, v, pp}, n, done));
        else KR_TRY(launch_ew(ctx, BicgPOp<false>{st, r, v, pp}, n, done));                                   // :126-142
        if (pc) { KR_TRY(pc_apply_dev(pc, pp, ph, done)); KR_TRY(launch_spmv(a, ph, v, 1, rhat, done)); }
        else KR_TRY(launch_spmv(a, pp, v, 1, rhat, done));                                        // :144-146 + (rhat,v)
        KR_TRY((reduce_then<1>(ctx, nt, ws.red, BicgAlphaLogic{lc})));
        KR_TRY(launch_ew(ctx, BicgSOp{st, r, v, s}, n, done));                                    // :166-188
        KR_TRY((reduce_then<1>(ctx, nt, ws.red, BicgSLogic{lc})));
        if (pc) { KR_TRY(pc_apply_dev(pc, s, sh, done)); KR_TRY(launch_spmv(a, sh, t, 2, s, done)); }
        else KR_TRY(launch_spmv(a, s, t, 2, s, done));                                            // :208-209 + (t,s),(t,t)
        KR_TRY((reduce_then<2>(ctx, nt, ws.red, BicgOmegaLogic{lc})));
        if (keep_in_cache(n)) KR_TRY(launch_ew_gated(ctx, BicgXROp<true>{st, pc ? ph : pp, pc ? sh : s, s, t, rhat, xw, r}, n, GateEarly{st}));
        else KR_TRY(launch_ew_gated(ctx, BicgXROp<false>{st, pc ? ph : pp, pc ? sh : s, s, t, rhat, xw, r}, n, GateEarly{st}));
        return reduce_then<2>(ctx, nt, ws.red, BicgEndLogic{lc});
    }
};

int32_t bicgstab_solve(kryst_vec_t bv, kryst_vec_t xv, const SolveIO& io, bool right_pc) {
    KR_ARG(io.a && io.params, "solve: null argument");
    BicgRun run(bv, xv, io, right_pc);
    return run.solve();
}

int32_t gmres_solve(kryst_vec_t bv, kryst_vec_t xv, const SolveIO& io);    // gmres.hip
int32_t cgs_solve(kryst_vec_t bv, kryst_vec_t xv, const SolveIO& io);      // cgs_tfqmr.hip
int32_t tfqmr_solve(kryst_vec_t bv, kryst_vec_t xv, const SolveIO& io);
SolverRun* make_cgs_run(kryst_vec_t b, kryst_vec_t x, const SolveIO& io);
SolverRun* make_tfqmr_run(kryst_vec_t b, kryst_vec_t x, const SolveIO& io);
int32_t fgmres_solve(kryst_vec_t bv, kryst_vec_t xv, const SolveIO& io, int32_t orthog, double haptol, int32_t preallocate);   // fgmres.hip

// ---- one checked reduction through freshly mapped mailboxes (dist.cpp: ipc_reduce_setup) before a solver relies on them: rank r sends
// r + 1 through the very kernel the inner products use, every rank must read P (P + 1) / 2, and the verdict is agreed through RCCL -- the
// mailbox path is on everywhere or nowhere.  What the set-up cannot see (a mapping that opens but does not carry system-scope stores
// between these two devices) ends here, under a short poll budget, not in a user's solve.  KRYST_IPC_SELFTEST=0 skips it.
struct IpcSelfTestLogic {
    static constexpr bool RUN_WHEN_DONE = false;
    LogicCtx c;
    __device__ void run(const double* red) const { c.st->rsq = red[0]; }
};
int32_t ipc_reduce_selftest(kryst_ctx_t ctx) {
    if (env_int("KRYST_IPC_SELFTEST", 1) == 0) return KRYST_OK;
    const int P = ctx->nranks;
    int64_t ok = 1;
    int32_t rc = ensure_partials(ctx, 1);
    if (rc != KRYST_OK) ok = 0;
    DevState* st = reinterpret_cast<DevState*>(ctx->d_scal);
    double* red = ctx->d_scal + 256;
    const double mine = (double)(ctx->rank + 1);
    if (ok && (hipMemsetAsync(ctx->d_scal, 0, sizeof(double) * 512, ctx->s_main) != hipSuccess ||
               hipMemcpyAsync(ctx->d_partials, &mine, sizeof mine, hipMemcpyHostToDevice, ctx->s_main) != hipSuccess ||
               hipStreamSynchronize(ctx->s_main) != hipSuccess)) { (void)hipGetLastError(); ok = 0; }
    if (ok) {
        const LogicCtx lc{st, nullptr, ctx->d_prog, red, 0.0, 1ll, 0, 0ll, 0};
        const IpcView v{ctx->ipc_mine, ctx->d_ipc_peers, ctx->d_ipc_epoch, ctx->rank, P, 1 << 24};     // (short by the solvers' standards -- 2^26 -- yet long enough for peers that time-slice one GPU in the tests)
        hipLaunchKernelGGL((fold_ipc_logic_kernel<1, IpcSelfTestLogic>), dim3(1), dim3(KR_F), 0, ctx->s_main,
                           ctx->d_partials, ctx->partials_cap, (int64_t)1, ctx->d_chunks, ctx->chunks_cap, fold_ticket(ctx), fold_err(ctx), red, IpcSelfTestLogic{lc}, v);
        DevState h;
        if (hipGetLastError() != hipSuccess || hipMemcpyAsync(&h, st, sizeof h, hipMemcpyDeviceToHost, ctx->s_main) != hipSuccess ||
            hipStreamSynchronize(ctx->s_main) != hipSuccess) { (void)hipGetLastError(); ok = 0; }
        else ok = (h.status == KRYST_OK && h.done == 0 && h.rsq == 0.5 * (double)P * (double)(P + 1)) ? 1 : 0;
    }
    ctx->h_prog->done = 0; ctx->h_prog->status = 0;                     // (a failed test went through LogicCtx::finish)
    (void)fold_gave_up(ctx);
    int64_t *d_s = nullptr, *d_r = nullptr;
    std::vector<int64_t> all((size_t)P, 0);
    rc = KRYST_OK;
    if (hipMalloc(&d_s, 8) != hipSuccess || hipMalloc(&d_r, sizeof(int64_t) * P) != hipSuccess) rc = KRYST_ERR_HIP;
    if (rc == KRYST_OK && (hipMemcpyAsync(d_s, &ok, 8, hipMemcpyHostToDevice, ctx->s_main) != hipSuccess || hipStreamSynchronize(ctx->s_main) != hipSuccess)) rc = KRYST_ERR_HIP;
    if (rc == KRYST_OK) rc = comm_all_gather_i64(ctx, d_s, d_r, 1, ctx->s_main);
    if (rc == KRYST_OK && (hipMemcpyAsync(all.data(), d_r, sizeof(int64_t) * P, hipMemcpyDeviceToHost, ctx->s_main) != hipSuccess || hipStreamSynchronize(ctx->s_main) != hipSuccess)) rc = KRYST_ERR_HIP;
    (void)hipFree(d_s); (void)hipFree(d_r);
    if (rc != KRYST_OK) { (void)hipGetLastError(); ctx->ipc_on = false; return rc; }
    for (int p = 0; p < P; ++p)
        if (all[(size_t)p] != 1) {
            ctx->ipc_on = false; ctx->ipc_failed = true;                // (the mailboxes stay mapped; the path is never switched on again)
            set_error("scalar all-reduce through mailboxes: the test reduction did not arrive intact on rank %d; the RCCL path stays in use", p);
            return KRYST_UNSUPPORTED;
        }
    return KRYST_OK;
}

}  // namespace kr

using namespace kr;

// host-slice entry points: upload b and the initial guess, solve on the device, download x
template <class F>
static int32_t host_solve(const double* b, double* x, int64_t n, const SolveIO& io, F f) {
    KR_ARG(io.a && b && x, "solve: null argument");
    KR_ARG(n == io.a->nrows, "solve: b.len() != nrows");
    kryst_vec_t bv = nullptr, xv = nullptr;
    KR_TRY(kryst_vec_create(io.a->ctx, n, &bv));
    int32_t rc = kryst_vec_create(io.a->ctx, n, &xv);
    if (rc == KRYST_OK) rc = kryst_vec_upload(bv, b, n);
    if (rc == KRYST_OK) rc = kryst_vec_upload(xv, x, n);
    if (rc == KRYST_OK) {
        rc = f(bv, xv, io);
        if (rc == KRYST_OK) { int32_t rd = kryst_vec_download(xv, x, n); if (rd != KRYST_OK) rc = rd; }
    }
    kryst_vec_destroy(bv); kryst_vec_destroy(xv);
    return rc;
}

#define IO_FROM_ARGS SolveIO io{a, pc, params, stats, hist, hist_cap, hist_len, monitor, user}

// A solve whose preconditioner had to switch kernels mid-way (pc.h: pc_health) is repeated once from the caller's x: on an
// error the solvers never touch x, like the reference (monitor callbacks then start over).
template <class F>
static int32_t retry_on_pc_fallback(kryst_pc_t pc, F f) {
    int32_t rc = f();
    if (rc == KRYST_SOLVE_ERROR && pc_fell_back(pc)) rc = f();
    return rc;
}

struct kryst_session_s { SolverRun* run; };

extern "C" {

int32_t kryst_session_begin(int32_t method, kryst_vec_t b, kryst_vec_t x, kryst_csr_t a, kryst_pc_t pc,
                            const kryst_params_t* params, kryst_session_t* out) {
    KR_ARG(a && params && b && x && out, "session_begin: null argument");
    SolveIO io{a, pc, params, nullptr, nullptr, 0, nullptr, nullptr, nullptr};
    SolverRun* run = nullptr;
    switch (method) {
        case 0: run = new CgRun(b, x, io); break;
        case 1: run = new PcgRun(b, x, io); break;
        case 2: run = new BicgRun(b, x, io, false); break;
        case 3: run = make_cgs_run(b, x, io); break;
        case 4: run = make_tfqmr_run(b, x, io); break;
        default: set_error("session_begin: unknown method %d", method); return KRYST_ERR_ARG;
    }
    const int32_t rc = run->begin();
    if (rc != KRYST_OK) { delete run; return rc; }
    *out = new kryst_session_s{run};
    return KRYST_OK;
}

int32_t kryst_session_step(kryst_session_t s, int64_t k) {
    KR_ARG(s && s->run && k >= 0, "session_step");
    return s->run->step(k);
}

int32_t kryst_session_end(kryst_session_t s, kryst_stats_t* stats, double* hist, int64_t hist_cap, int64_t* hist_len) {
    KR_ARG(s && s->run, "session_end");
    s->run->io.stats = stats; s->run->io.hist = hist; s->run->io.hist_cap = hist_cap; s->run->io.hist_len = hist_len;
    const int32_t rc = s->run->end();
    delete s->run;
    delete s;
    return rc;
}

int32_t kryst_cg_solve_dev(kryst_vec_t b, kryst_vec_t x, KRYST_SOLVE_ARGS) { IO_FROM_ARGS; return cg_solve(b, x, io); }
int32_t kryst_pcg_solve_dev(kryst_vec_t b, kryst_vec_t x, KRYST_SOLVE_ARGS) { IO_FROM_ARGS; return retry_on_pc_fallback(pc, [&] { return pcg_solve(b, x, io); }); }
int32_t kryst_gmres_solve_dev(kryst_vec_t b, kryst_vec_t x, KRYST_SOLVE_ARGS) { IO_FROM_ARGS; return retry_on_pc_fallback(pc, [&] { return gmres_solve(b, x, io); }); }
int32_t kryst_cgs_solve_dev(kryst_vec_t b, kryst_vec_t x, KRYST_SOLVE_ARGS) { IO_FROM_ARGS; return cgs_solve(b, x, io); }
int32_t kryst_tfqmr_solve_dev(kryst_vec_t b, kryst_vec_t x, KRYST_SOLVE_ARGS) { IO_FROM_ARGS; return tfqmr_solve(b, x, io); }
int32_t kryst_cgs_solve(const double* b, double* x, int64_t n, KRYST_SOLVE_ARGS) {
    IO_FROM_ARGS; return host_solve(b, x, n, io, [](kryst_vec_t bv, kryst_vec_t xv, const SolveIO& i) { return cgs_solve(bv, xv, i); });
}
int32_t kryst_tfqmr_solve(const double* b, double* x, int64_t n, KRYST_SOLVE_ARGS) {
    IO_FROM_ARGS; return host_solve(b, x, n, io, [](kryst_vec_t bv, kryst_vec_t xv, const SolveIO& i) { return tfqmr_solve(bv, xv, i); });
}
int32_t kryst_fgmres_solve_dev(kryst_vec_t b, kryst_vec_t x, int32_t orthog, double haptol, int32_t preallocate, KRYST_SOLVE_ARGS) {
    IO_FROM_ARGS; return retry_on_pc_fallback(pc, [&] { return fgmres_solve(b, x, io, orthog, haptol, preallocate); });
}
int32_t kryst_fgmres_solve(const double* b, double* x, int64_t n, int32_t orthog, double haptol, int32_t preallocate, KRYST_SOLVE_ARGS) {
    IO_FROM_ARGS;
    return host_solve(b, x, n, io, [=](kryst_vec_t bv, kryst_vec_t xv, const SolveIO& i) { return retry_on_pc_fallback(i.pc, [&] { return fgmres_solve(bv, xv, i, orthog, haptol, preallocate); }); });
}
int32_t kryst_bicgstab_solve_dev(kryst_vec_t b, kryst_vec_t x, KRYST_SOLVE_ARGS) { IO_FROM_ARGS; return bicgstab_solve(b, x, io, false); }
int32_t kryst_bicgstab_rpc_solve_dev(kryst_vec_t b, kryst_vec_t x, KRYST_SOLVE_ARGS) { IO_FROM_ARGS; return retry_on_pc_fallback(pc, [&] { return bicgstab_solve(b, x, io, true); }); }

int32_t kryst_cg_solve(const double* b, double* x, int64_t n, KRYST_SOLVE_ARGS) {
    IO_FROM_ARGS; return host_solve(b, x, n, io, [](kryst_vec_t bv, kryst_vec_t xv, const SolveIO& i) { return cg_solve(bv, xv, i); });
}
int32_t kryst_pcg_solve(const double* b, double* x, int64_t n, KRYST_SOLVE_ARGS) {
    IO_FROM_ARGS; return host_solve(b, x, n, io, [](kryst_vec_t bv, kryst_vec_t xv, const SolveIO& i) { return retry_on_pc_fallback(i.pc, [&] { return pcg_solve(bv, xv, i); }); });
}
int32_t kryst_gmres_solve(const double* b, double* x, int64_t n, KRYST_SOLVE_ARGS) {
    IO_FROM_ARGS; return host_solve(b, x, n, io, [](kryst_vec_t bv, kryst_vec_t xv, const SolveIO& i) { return retry_on_pc_fallback(i.pc, [&] { return gmres_solve(bv, xv, i); }); });
}
int32_t kryst_bicgstab_solve(const double* b, double* x, int64_t n, KRYST_SOLVE_ARGS) {
    IO_FROM_ARGS; return host_solve(b, x, n, io, [](kryst_vec_t bv, kryst_vec_t xv, const SolveIO& i) { return bicgstab_solve(bv, xv, i, false); });
}

}  // extern "C"
