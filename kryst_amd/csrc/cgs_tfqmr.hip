// Device-resident CGS and TFQMR: CgsSolver::solve (src/solver/cgs.rs:58-135) and TfqmrSolver::solve
// (src/solver/tfqmr.rs:64-221), operation by operation.  Both ignore the preconditioner argument like the reference
// (cgs.rs:59, tfqmr.rs:66); TFQMR also discards the initial guess (tfqmr.rs:72).
//
// HBM passes per iteration (n-word vectors, SpMV aside):
//   CGS    3 fused kernels, 15 words  (reference: 8 loops + 3 clones, 29 words)
//   TFQMR  3 fused kernels, 16 words  (reference: 12 loops + 1 clone, 40 words; its `w` vector and the residual
//          r - alpha*A(u+q) are never read again -- only the norm of the latter is -- so neither is stored)
#include "solver_run.h"

namespace kr {

// =================================================================== CGS
struct CgsDirOp {                    // i == 1: u = r, p = u (cgs.rs:83-86); else u = r + beta q, p = u + beta (q + beta p) (:87-99)
    static constexpr int NQ = 0; static constexpr const char* TAG = "CgsDir";
    const DevState* st; int first; const double* r; const double* q; double* u; double* p;
    __device__ __forceinline__ void pair(int64_t i, bool, bool, double (&)[1]) const {
        const d2 rr = ld2(r, i);
        if (first) { st2(u, i, rr.a, rr.b); st2(p, i, rr.a, rr.b); return; }
        const double be = st->beta;
        const d2 qq = ld2(q, i), pp = ld2(p, i);
        const double u0 = rr.a + be * qq.a, u1 = rr.b + be * qq.b;
        st2(u, i, u0, u1);
        st2(p, i, u0 + be * (qq.a + be * pp.a), u1 + be * (qq.b + be * pp.b));
    }
};
struct CgsQxOp {                     // q = u - alpha v (:107-109) ; x += alpha (u + q) (:111-113) ; upq = u + q (:115-118)
    static constexpr int NQ = 0; static constexpr const char* TAG = "CgsQx";
    const DevState* st; const double* u; const double* v; double* q; double* x; double* upq;
    __device__ __forceinline__ void pair(int64_t i, bool, bool, double (&)[1]) const {
        const double al = st->alpha;
        const d2 uu = ld2(u, i), vv = ld2(v, i), xx = ld2(x, i);
        const double q0 = uu.a - al * vv.a, q1 = uu.b - al * vv.b;
        const double s0 = uu.a + q0, s1 = uu.b + q1;
        st2(q, i, q0, q1);
        st2(x, i, xx.a + al * s0, xx.b + al * s1);
        st2(upq, i, s0, s1);
    }
};
struct CgsROp {                      // r = r - alpha w (:121-123) ; partials (r,r) (:124) and (r_tld, r) (:133)
    static constexpr int NQ = 2; static constexpr const char* TAG = "CgsR";
    const DevState* st; const double* w; const double* rt; double* r;
    __device__ __forceinline__ void pair(int64_t i, bool in0, bool in1, double (&acc)[2]) const {
        const double al = st->alpha;
        const d2 rr = ld2(r, i), ww = ld2(w, i), tt = ld2(rt, i);
        const double r0 = rr.a - al * ww.a, r1 = rr.b - al * ww.b;
        st2(r, i, r0, r1);
        if (in0) { acc[0] = acc[0] + r0 * r0; acc[1] = acc[1] + tt.a * r0; }
        if (in1) { acc[0] = acc[0] + r1 * r1; acc[1] = acc[1] + tt.b * r1; }
    }
};
struct CgsInitLogic {                // cgs.rs:74-82 ; red0 = (r,r) = (r_tld,r)
    static constexpr bool RUN_WHEN_DONE = false;
    LogicCtx c;
    __device__ void run(const double* red) const {
        DevState* st = c.st;
        st->rho = red[0]; st->rho_prev = 0.0;                           // :74-75
        st->res0 = dsqrt(red[0]);                                       // :76
        st->iterations = 0; st->final_residual = st->res0; st->converged = 0; st->iter = 0;   // :77
        if (c.max_iters <= 0) { c.finish(KRYST_OK); return; }
        if (fabs(st->rho) < DBL_EPSILON) c.finish(KRYST_OK);            // :80-82 break at i = 1
    }
};
struct CgsAlphaLogic {               // :105 ; red0 = (r_tld, v); no guard on the denominator
    static constexpr bool RUN_WHEN_DONE = false;
    LogicCtx c;
    __device__ void run(const double* red) const { c.st->alpha = c.st->rho / red[0]; }
};
struct CgsEndLogic {                 // :124-133 then the head of the next iteration :80-88
    static constexpr bool RUN_WHEN_DONE = false;
    LogicCtx c;
    __device__ void run(const double* red) const {
        DevState* st = c.st;
        const long long i = st->iter + 1;
        const double res_norm = dsqrt(red[0]);
        c.push(res_norm);                                               // addition: the reference keeps no history
        st->iter = i;
        if (c.check(res_norm, st->res0, i)) { c.finish(KRYST_OK); return; }   // :126-131 (stop implies s.converged)
        st->rho_prev = st->rho; st->rho = red[1];                       // :132-133
        if (i >= c.max_iters) { c.finish(KRYST_OK); return; }           // unreachable (check stops at the cap); kept for safety
        if (fabs(st->rho) < DBL_EPSILON) { c.finish(KRYST_OK); return; }      // :80-82
        st->beta = st->rho / st->rho_prev;                              // :88
    }
};

struct CgsRun : SolverRun {
    using SolverRun::SolverRun;
    double *r = nullptr, *rt = nullptr, *pp = nullptr, *q = nullptr, *u = nullptr, *v = nullptr, *upq = nullptr;
    int32_t begin() override {
        KR_TRY(solve_args_check(io, bv, xv));
        KR_TRY(common_begin(prm.max_iters + 2, 7));
        pc = nullptr;                                                                             // cgs.rs:59
        KR_TRY(ws.vec(&r)); KR_TRY(ws.vec(&rt)); KR_TRY(ws.vec(&pp)); KR_TRY(ws.vec(&q)); KR_TRY(ws.vec(&u)); KR_TRY(ws.vec(&v));
        KR_TRY(ws.vec(&upq));
        KR_TRY(residual_dot(a, bv->d, xw, r, v, nullptr));                                        // :64-69, :74, :76
        KR_HIP(hipMemcpyAsync(rt, r, padded_bytes(n), hipMemcpyDeviceToDevice, ctx->s_main));      // :70
        return reduce_then<1>(ctx, nt, ws.red, CgsInitLogic{lc});
    }
    int32_t iterate(int64_t i) override {
        const DevState* st = ws.st;
        KR_TRY(launch_ew(ctx, CgsDirOp{st, i == 1 ? 1 : 0, r, q, u, pp}, n, done));               // :83-99
        KR_TRY(launch_spmv(a, pp, v, 1, rt, done));                                               // :101-103 + (r_tld, v)
        KR_TRY((reduce_then<1>(ctx, nt, ws.red, CgsAlphaLogic{lc})));
        KR_TRY(launch_ew(ctx, CgsQxOp{st, u, v, q, xw, upq}, n, done));                           // :107-118
        KR_TRY(launch_spmv(a, upq, v, 0, nullptr, done));                                         // :119-120 (w reuses v's storage)
        KR_TRY(launch_ew(ctx, CgsROp{st, v, rt, r}, n, done));                                    // :121-124, :133
        return reduce_then<2>(ctx, nt, ws.red, CgsEndLogic{lc});
    }
};

int32_t cgs_solve(kryst_vec_t bv, kryst_vec_t xv, const SolveIO& io) {
    KR_ARG(io.a && io.params, "solve: null argument");
    CgsRun run(bv, xv, io);
    return run.solve();
}

// =================================================================== TFQMR
struct TfState {                     // device, next to DevState
    double dpold, uu, qq, rho_new, psi_old, eta_old;
    double eta0, cf0, eta1, cf1;
    int stop0;                       // the m = 0 substep returned: skip m = 1 (tfqmr.rs:191-196)
};
struct TfUqtOp {                     // u = r - alpha v (:131-133) ; q = u - alpha v (:136-139) ; t = u + q (:142-145)
    static constexpr int NQ = 3; static constexpr const char* TAG = "TfUqt";     // partials (u,u) [||r|| of a later early return, :117/:124], (q,q) (:160), (r_tld,u) (:204)
    const DevState* st; const double* r; const double* v; const double* rt; double* u; double* q; double* t;
    __device__ __forceinline__ void pair(int64_t i, bool in0, bool in1, double (&acc)[3]) const {
        const double al = st->alpha;
        const d2 rr = ld2(r, i), vv = ld2(v, i), tt = ld2(rt, i);
        const double u0 = rr.a - al * vv.a, u1 = rr.b - al * vv.b;
        const double q0 = u0 - al * vv.a, q1 = u1 - al * vv.b;
        st2(u, i, u0, u1); st2(q, i, q0, q1); st2(t, i, u0 + q0, u1 + q1);
        if (in0) { acc[0] = acc[0] + u0 * u0; acc[1] = acc[1] + q0 * q0; acc[2] = acc[2] + u0 * tt.a; }
        if (in1) { acc[0] = acc[0] + u1 * u1; acc[1] = acc[1] + q1 * q1; acc[2] = acc[2] + u1 * tt.b; }
    }
};
struct TfResNormOp {                 // partial ||r - alpha A(u+q)||^2 (:149-152); the vector itself is dead (r = u at :203)
    static constexpr int NQ = 1; static constexpr const char* TAG = "TfResNorm"; static constexpr int BPC = 4;
    const DevState* st; const double* r; const double* au;
    __device__ __forceinline__ void pair(int64_t i, bool in0, bool in1, double (&acc)[1]) const {
        const double al = st->alpha;
        const d2 rr = ld2(r, i), aa = ld2(au, i);
        const double r0 = rr.a - al * aa.a, r1 = rr.b - al * aa.b;
        if (in0) acc[0] = acc[0] + r0 * r0;
        if (in1) acc[0] = acc[0] + r1 * r1;
    }
};
struct TfXdyOp {                     // both substeps of :175-182, then y = u + beta (q + beta y) (:210)
    static constexpr int NQ = 0; static constexpr const char* TAG = "TfXdy";
    const DevState* st; const TfState* tf; const double* u; const double* q; double* d; double* x; double* y;
    __device__ __forceinline__ void pair(int64_t i, bool, bool, double (&)[1]) const {
        const d2 uu = ld2(u, i), dd = ld2(d, i), xx = ld2(x, i);
        double d0 = uu.a + tf->cf0 * dd.a, d1 = uu.b + tf->cf0 * dd.b;        // m = 0: D = U + cf D
        double x0 = xx.a + tf->eta0 * d0, x1 = xx.b + tf->eta0 * d1;          //        x = x + eta D
        if (tf->stop0) { st2(d, i, d0, d1); st2(x, i, x0, x1); return; }
        const d2 qq = ld2(q, i);
        d0 = qq.a + tf->cf1 * d0; d1 = qq.b + tf->cf1 * d1;                   // m = 1: D = Q + cf D
        x0 = x0 + tf->eta1 * d0; x1 = x1 + tf->eta1 * d1;
        st2(d, i, d0, d1); st2(x, i, x0, x1);
        if (st->early) return;                                                // returned at m = 1: y is dead
        const double be = st->beta;
        const d2 yy = ld2(y, i);
        st2(y, i, uu.a + be * (qq.a + be * yy.a), uu.b + be * (qq.b + be * yy.b));
    }
};
struct TfInitLogic {                 // tfqmr.rs:80-107 ; red0 = (r, r_tld) = (b,b)
    static constexpr bool RUN_WHEN_DONE = false;
    LogicCtx c; TfState* tf;
    __device__ void run(const double* red) const {
        DevState* st = c.st;
        st->rho = red[0];
        const double tau = dsqrt(red[0]);
        st->iter = 0; st->iterations = 0; st->res0 = tau; st->final_residual = tau; st->converged = 0;
        tf->uu = red[0];                                                // ||r||^2 of the current r
        tf->dpold = tau; tf->psi_old = 0.0; tf->eta_old = 0.0; tf->stop0 = 0;   // :98-99, :107
        if (st->rho == 0.0) { st->converged = 1; c.finish(KRYST_OK); return; }   // :81-83 (final_residual = ||r||)
        if (tau == 0.0) { st->final_residual = 0.0; st->converged = 1; c.finish(KRYST_OK); return; }   // :103-105
        if (c.max_iters <= 0) c.finish(KRYST_OK);                       // :215-217 with an empty loop: ||r||, iterations = 0
    }
};
struct TfAlphaLogic {                // :115-128 ; red0 = (r_tld, v)
    static constexpr bool RUN_WHEN_DONE = false;
    LogicCtx c; TfState* tf;
    __device__ void run(const double* red) const {
        DevState* st = c.st;
        const long long k = st->iter + 1;
        const double sigma = red[0];
        bool bad = (sigma == 0.0 || !isfinite(sigma));
        double alpha = 0.0;
        if (!bad) { alpha = st->rho / sigma; bad = (alpha == 0.0 || !isfinite(alpha)); }
        if (bad) {
            st->final_residual = dsqrt(tf->uu); st->iterations = k; st->converged = 0;
            c.finish(KRYST_OK);
            return;
        }
        st->alpha = alpha;
    }
};
struct TfStoreLogic {                // keeps (u,u), (q,q), (r_tld,u) for the steps after the second SpMV
    static constexpr bool RUN_WHEN_DONE = false;
    LogicCtx c; TfState* tf;
    __device__ void run(const double* red) const { tf->uu = red[0]; tf->qq = red[1]; tf->rho_new = red[2]; }
};
struct TfStepLogic {                 // :152-212 scalars ; red0 = ||r - alpha A(u+q)||^2
    static constexpr bool RUN_WHEN_DONE = false;
    LogicCtx c; TfState* tf;
    __device__ void run(const double* red) const {
        DevState* st = c.st;
        const long long k = st->iter + 1;
        const double alpha = st->alpha;
        const double dp = dsqrt(red[0]);                                // :152
        const double tau_m0 = dsqrt(dp * tf->dpold);                    // :153
        double tau_local = tau_m0;
        tf->stop0 = 0;
        for (int m = 0; m < 2; ++m) {                                   // :156
            const double norm_u_m = (m == 0) ? dp : dsqrt(tf->qq);      // :157-161
            const double tau_for_m = (m == 0) ? tau_m0 : tau_local;
            const double psi = norm_u_m / tau_for_m;                    // :165
            const double c_m = 1.0 / dsqrt(1.0 + psi * psi);            // :166
            const double eta = c_m * c_m * alpha;                       // :167
            const double cf = (alpha == 0.0 || k == 1) ? 0.0 : tf->psi_old * tf->psi_old * tf->eta_old / alpha;   // :170-174
            if (m == 0) { tf->eta0 = eta; tf->cf0 = cf; } else { tf->eta1 = eta; tf->cf1 = cf; }
            const double dpest = dsqrt((double)(2 * k + m + 2)) * tau_for_m;   // :185
            c.push(dpest);                                              // addition: the reference keeps no history
            const bool stop = c.check(dpest, st->res0, k);              // :186-187
            tf->psi_old = psi; tf->eta_old = eta;                       // :188-189
            tau_local = tau_for_m * psi * c_m;                          // :190
            if (stop) {                                                 // :191-196
                st->final_residual = dpest; st->iterations = k; st->converged = 1;
                if (m == 0) tf->stop0 = 1;
                st->early = 1;                                          // the x update of this substep is still to run
                st->iter = k;
                c.finish(KRYST_OK);
                return;
            }
        }
        st->beta = tf->rho_new / st->rho;                               // :204-205 (r = u, so (r_tld, r) = (r_tld, u))
        st->rho = tf->rho_new;
        tf->dpold = dp;                                                 // :212
        st->iter = k;
    }
};

struct TfqmrRun : SolverRun {
    using SolverRun::SolverRun;
    double *rbuf[2] = {nullptr, nullptr}, *rt = nullptr, *v = nullptr, *y = nullptr, *q = nullptr, *t = nullptr, *d = nullptr;
    TfState* tf = nullptr;
    int32_t begin() override {
        KR_TRY(solve_args_check(io, bv, xv));
        KR_TRY(common_begin(2 * prm.max_iters + 2, 8));
        pc = nullptr;                                                                             // tfqmr.rs:66
        tf = reinterpret_cast<TfState*>(ctx->d_scal + 128);         // d_scal: DevState at 0, TfState at +128, red at +256 doubles
        KR_TRY(ws.vec(&rbuf[0])); KR_TRY(ws.vec(&rbuf[1])); KR_TRY(ws.vec(&rt)); KR_TRY(ws.vec(&v)); KR_TRY(ws.vec(&y));
        KR_TRY(ws.vec(&q)); KR_TRY(ws.vec(&t)); KR_TRY(ws.vec(&d));
        KR_HIP(hipMemsetAsync(xw, 0, padded_bytes(n), ctx->s_main));                               // :72  x = 0
        KR_HIP(hipMemcpyAsync(rbuf[0], bv->d, padded_bytes(n), hipMemcpyDeviceToDevice, ctx->s_main));   // :75
        KR_HIP(hipMemcpyAsync(rt, bv->d, padded_bytes(n), hipMemcpyDeviceToDevice, ctx->s_main));        // :77
        KR_HIP(hipMemcpyAsync(y, bv->d, padded_bytes(n), hipMemcpyDeviceToDevice, ctx->s_main));         // :95
        KR_TRY(launch_ew(ctx, DotOneOp{rbuf[0], rt}, n, nullptr));                                // :80, :100
        return reduce_then<1>(ctx, nt, ws.red, TfInitLogic{lc, tf});
    }
    int32_t iterate(int64_t k) override {
        const DevState* st = ws.st;
        // `r.clone_from(&u)` (:203) is a buffer swap: iteration k reads r from rbuf[(k-1)&1] and writes u to the other
        double* r = rbuf[(k - 1) & 1]; double* u = rbuf[k & 1];
        KR_TRY(launch_spmv(a, y, v, 1, rt, done));                                                // :110-115
        KR_TRY((reduce_then<1>(ctx, nt, ws.red, TfAlphaLogic{lc, tf})));
        KR_TRY(launch_ew(ctx, TfUqtOp{st, r, v, rt, u, q, t}, n, done));                          // :131-145
        KR_TRY((reduce_then<3>(ctx, nt, ws.red, TfStoreLogic{lc, tf})));
        KR_TRY(launch_spmv(a, t, v, 0, nullptr, done));                                           // :146-147 (au reuses v's storage)
        KR_TRY(launch_ew(ctx, TfResNormOp{st, r, v}, n, done));                                   // :149-152
        KR_TRY((reduce_then<1>(ctx, nt, ws.red, TfStepLogic{lc, tf})));
        KR_TRY(launch_ew_gated(ctx, TfXdyOp{st, tf, u, q, d, xw, y}, n, GateEarly{st}));
        hipLaunchKernelGGL((logic_kernel<ClearEarlyLogic>), dim3(1), dim3(64), 0, ctx->s_main, ws.red, ClearEarlyLogic{lc});
        KR_HIP(hipGetLastError());
        return KRYST_OK;
    }
};

SolverRun* make_cgs_run(kryst_vec_t b, kryst_vec_t x, const SolveIO& io) { return new CgsRun(b, x, io); }
SolverRun* make_tfqmr_run(kryst_vec_t b, kryst_vec_t x, const SolveIO& io) { return new TfqmrRun(b, x, io); }

int32_t tfqmr_solve(kryst_vec_t bv, kryst_vec_t xv, const SolveIO& io) {
    KR_ARG(io.a && io.params, "solve: null argument");
    TfqmrRun run(bv, xv, io);
    return run.solve();
}

}  // namespace kr
