// Device-resident restarted GMRES(m): GmresSolver::solve (src/solver/gmres.rs:216-402) with its helpers
// arnoldi (:65-105), apply_givens_and_update_g (:154-176) and back_substitution (:180-192), restated literally
// -- including the non-standard Left variant (orthogonalisation against Z[0..=j] with Z[0] = M^-1 v0, x updated
// with V), the branch asymmetries on happy breakdown and the epsilon = 1e-14 guards.
//
// Hessenberg matrix, Givens rotations and g live in device memory and are updated by one-thread logic kernels,
// so the 2(j+1) dependent dot -> axpy links of the double modified Gram-Schmidt sweep never touch the host.
// Each link is ONE pass over HBM:  z <- z - h_i B_i  fused with the next link's dot (z, B_{i+1})  (4n words
// instead of the reference's 5n); the last link fuses ||z||^2.
#include "solver_common.h"

namespace kr {

struct GmState {                     // device
    int cyc_stop;                    // leave the Arnoldi loop of this cycle
    int happy;
    int m;
    int side;                        // 0 none, 1 left, 2 right, 3 textbook left (labelled extension)
    long long iteration;
    double r0_norm, beta, hcur, hj1;
    double res0_in;                  // the norm the in-cycle convergence test divides by: res0, or (side 3) ||M^-1 r0|| of the first cycle
};

struct GmPtrs { GmState* gs; double* h; double* g; double* cs; double* sn; double* y; int restart; };

// ---- vector ops
struct DivOp {                       // out = in / s      (gmres.rs:242,253,304 `ri / r0_norm`, `zi / h[j+1][j]`)
    static constexpr int NQ = 0; static constexpr const char* TAG = "Div";
    const double* s; const double* in; double* out;
    __device__ __forceinline__ void pair(int64_t i, bool, bool, double (&)[1]) const {
        const double d = *s;
        const d2 a = ld2(in, i);
        st2(out, i, a.a / d, a.b / d);
    }
};
template <bool KEEP>
struct MgsLinkOp {                   // z = z - h*Bi (gmres.rs:85-87) ; partial z.Bnext (the next link's dot, :84/:91)
    static constexpr int NQ = 1; static constexpr const char* TAG = "MgsLink";         // bnext == nullptr: partial z.z of the UPDATED z (h[j+1][j] = ||z||, :97)
    const double* h; const double* bi; const double* bnext; double* z;
    __device__ __forceinline__ void pair(int64_t i, bool in0, bool in1, double (&acc)[1]) const {
        const double hv = *h;
        // z is read again by the very next link and Bnext is its Bi: cacheable accesses for those two, streaming for Bi
        // when they can survive in the memory-side cache (GMRES(30): 256^3 310 -> 336 it/s, 128^3 1377 -> 1495)
        const d2 zz = ld2_sel<KEEP>(z, i), b = ld2(bi, i);
        const double z0 = zz.a - hv * b.a, z1 = zz.b - hv * b.b;
        st2_sel<KEEP>(z, i, z0, z1);
        d2 nx{z0, z1};
        if (bnext) nx = ld2_sel<KEEP>(bnext, i);
        if (in0) acc[0] = acc[0] + z0 * nx.a;
        if (in1) acc[0] = acc[0] + z1 * nx.b;
    }
};
struct GmUpdateOp {                  // x += sum_j y[j]*U[j], j ascending per element (gmres.rs:362-386)
    static constexpr int NQ = 0; static constexpr const char* TAG = "GmUpdate";
    const GmState* gs; const double* y; double* const* u; double* x;
    __device__ __forceinline__ void pair(int64_t i, bool, bool, double (&)[1]) const {
        const int m = gs->m;
        d2 xx = ld2(x, i);
        for (int j0 = 0; j0 < m; j0 += 8) {                  // 8 basis vectors in flight; the sum keeps its ascending order
            d2 uu[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) uu[k] = ld2(u[min(j0 + k, m - 1)], i);
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (j0 + k < m) { const double yj = y[j0 + k]; xx.a = xx.a + yj * uu[k].a; xx.b = xx.b + yj * uu[k].b; }
        }
        st2(x, i, xx.a, xx.b);
    }
};

// work inside the Arnoldi loop is gated on done || cyc_stop
struct GateCycle {
    const DevState* st; const GmState* gs;
    __device__ __forceinline__ bool skip() const { return st->done || gs->cyc_stop; }
};
// the Gram-Schmidt links (3 reads + 1 write + a reduction per tile) want 6 workgroups per CU WHILE z lives in the memory-side cache: GMRES(30)
// 256^3 244 -> 309 it/s with 4 (round 1), 347.7 -> 356.5 with 6 (round 3, tools/solver_ab.py: 2 / 3 / 4 / 5 / 6 / 8 per CU = 250 / 319 / 348 / 356 /
// 357 / 339 it/s).  Beyond the cache every word comes from HBM and the streams' rule holds (ew.h: a narrow moving window keeps DRAM pages open) --
// round 5, tools/fgmres_only.py: 512^3 38.0 / 40.1 / 41.0 / 33.2 it/s with 6 / 4 / 3 / 2 per CU, 384^3 94.9 / 97.6 / 97.8 with 6 / 4 / 3,
// 320^3 165.3 / 169.6 / 166.3: 6 in the cache regime, 3 for vectors beyond 768 MiB, 4 in between.
template <class Op>
static int32_t launch_iter(kryst_ctx_t ctx, const Op& op, int64_t n, const DevState* st, const GmState* gs) {
    static const int forced = [] { const char* e = getenv("KRYST_GMRES_BLOCKS_PER_CU"); return e ? atoi(e) : 0; }();
    const int bpc = forced > 0 ? forced : keep_in_cache(n) ? 6 : n * 8 > (768ll << 20) ? 3 : 4;
    return launch_ew_gated(ctx, op, n, GateCycle{st, gs}, bpc);
}
// "gate" kernel: copies done||cyc_stop into one int so that launch_spmv / pc_apply_dev can use their `done` hook
__global__ void gate_kernel(const DevState* st, const GmState* gs, int* gate) { *gate = (st->done || gs->cyc_stop) ? 1 : 0; }

// ---- logic
#define HH(i, k) P.h[(i) * P.restart + (k)]
struct GmInitLogic {                 // gmres.rs:227-233 ; red0 = (r0,r0)
    static constexpr bool RUN_WHEN_DONE = false;
    LogicCtx c; GmPtrs P; int side;
    __device__ void run(const double* red) const {
        DevState* st = c.st;
        const double beta = dsqrt(red[0]);
        P.gs->beta = beta; st->res0 = beta; P.gs->res0_in = beta;
        st->iterations = 0; st->final_residual = beta; st->converged = 0;
        P.gs->iteration = 0; P.gs->side = side;
        if (c.max_iters <= 0 || P.restart <= 0) c.finish(KRYST_OK);     // n_outer == 0 (:231,:234)
    }
};
struct GmCycleLogic {                // gmres.rs:238, :268-275 ; for Right: red0 = (z0,z0) (:252,:259)
    static constexpr bool RUN_WHEN_DONE = false;
    LogicCtx c; GmPtrs P; int right;
    __device__ void run(const double* red) const {
        GmState* gs = P.gs;
        double r0_norm = gs->beta;                                      // :238
        if (right) { r0_norm = dsqrt(red[0]); gs->beta = r0_norm; }     // :252, :259
        if (right == 2 && gs->iteration == 0) gs->res0_in = r0_norm;    // side 3: the first cycle's ||M^-1 r0||
        gs->r0_norm = r0_norm;
        for (int k = 0; k < (P.restart + 1) * P.restart; ++k) P.h[k] = 0.0;
        for (int k = 0; k <= P.restart; ++k) P.g[k] = 0.0;
        P.g[0] = r0_norm;                                               // :270
        for (int k = 0; k < P.restart; ++k) { P.cs[k] = 0.0; P.sn[k] = 0.0; P.y[k] = 0.0; }
        gs->m = 0; gs->happy = 0; gs->cyc_stop = 0;
    }
};
struct GmStepBeginLogic {            // gmres.rs:277 iteration += 1
    static constexpr bool RUN_WHEN_DONE = false;
    LogicCtx c; GmPtrs P;
    __device__ void run(const double*) const { if (!P.gs->cyc_stop) P.gs->iteration = P.gs->iteration + 1; }
};
struct GmHLogic {                    // first sweep: h[i][j] = dot (:84) ; second: h[i][j] += tmp (:91-92)
    static constexpr bool RUN_WHEN_DONE = false;
    LogicCtx c; GmPtrs P; int i, j, second;
    __device__ void run(const double* red) const {
        if (P.gs->cyc_stop) return;
        const double d = red[0];
        if (second) HH(i, j) = HH(i, j) + d; else HH(i, j) = d;
        P.gs->hcur = d;
    }
};
struct GmNormLogic {                 // gmres.rs:97-101 / :299-303 / :331-335 then :347-354
    static constexpr bool RUN_WHEN_DONE = false;
    LogicCtx c; GmPtrs P; int j;
    __device__ void run(const double* red) const {
        GmState* gs = P.gs;
        if (gs->cyc_stop) return;
        const double eps = 1e-14;                                       // :233
        const double hj1 = dsqrt(red[0]);
        HH(j + 1, j) = hj1; gs->hj1 = hj1;
        if (fabs(hj1) < eps) {
            gs->happy = 1;
            if (gs->side == 1 || gs->side == 2) { gs->cyc_stop = 1; return; }   // :300-303 / :332-335: break BEFORE givens, m unchanged (side 3: as arnoldi)
        }
        // apply_givens_and_update_g (:154-176)
        for (int i = 0; i < j; ++i) {
            const double temp = P.cs[i] * HH(i, j) + P.sn[i] * HH(i + 1, j);
            HH(i + 1, j) = -P.sn[i] * HH(i, j) + P.cs[i] * HH(i + 1, j);
            HH(i, j) = temp;
        }
        const double h_kk = HH(j, j), h_k1k = HH(j + 1, j);
        const double r = dsqrt(h_kk * h_kk + h_k1k * h_k1k);
        if (fabs(r) < eps) { P.cs[j] = 1.0; P.sn[j] = 0.0; }
        else { P.cs[j] = h_kk / r; P.sn[j] = h_k1k / r; }
        HH(j, j) = P.cs[j] * h_kk + P.sn[j] * h_k1k;
        HH(j + 1, j) = 0.0;
        const double temp = P.cs[j] * P.g[j] + P.sn[j] * P.g[j + 1];
        P.g[j + 1] = -P.sn[j] * P.g[j] + P.cs[j] * P.g[j + 1];
        P.g[j] = temp;
        const double res_norm = fabs(P.g[j + 1]);                       // :348
        const bool conv = c.check(res_norm, gs->res0_in, gs->iteration);   // :349-350 (res0_in == res0 unless side 3)
        c.push(res_norm);                                               // addition: the reference keeps no GMRES history
        gs->m = j + 1;                                                  // :351
        if (conv || gs->happy) gs->cyc_stop = 1;                        // :352-354
    }
};
struct GmBackLogic {                 // back_substitution (:180-192) on the leading m x m block
    static constexpr bool RUN_WHEN_DONE = false;
    LogicCtx c; GmPtrs P;
    __device__ void run(const double*) const {
        const double eps = 1e-14;
        const int m = P.gs->m;
        for (int i = m - 1; i >= 0; --i) {
            double yi = P.g[i];
            for (int k = i + 1; k < m; ++k) yi = yi - HH(i, k) * P.y[k];
            if (fabs(HH(i, i)) > eps) yi = yi / HH(i, i); else yi = 0.0;
            P.y[i] = yi;
        }
    }
};
struct GmCycleEndLogic {             // gmres.rs:392-398 ; red0 = (r0,r0) of the true residual
    static constexpr bool RUN_WHEN_DONE = false;
    LogicCtx c; GmPtrs P;
    __device__ void run(const double* red) const {
        DevState* st = c.st;
        const double beta = dsqrt(red[0]);
        P.gs->beta = beta;
        st->final_residual = beta;                                      // :394
        st->converged = (beta < c.tol * st->res0) ? 1 : 0;              // :395
        st->iter = P.gs->iteration;
        if (st->converged || P.gs->iteration >= c.max_iters) c.finish(KRYST_OK);   // :396-398
    }
};
#undef HH

template <class L>
static int32_t logic_only(kryst_ctx_t ctx, const double* red, const L& l) {
    hipLaunchKernelGGL((logic_kernel<L>), dim3(1), dim3(64), 0, ctx->s_main, red, l);
    KR_HIP(hipGetLastError());
    return KRYST_OK;
}

int32_t gmres_solve(kryst_vec_t bv, kryst_vec_t xv, const SolveIO& io) {
    const EnvFreeze knobs;                    // the tuning knobs are read once per solve, not per launch
    KR_ARG(io.a && io.params && bv && xv, "solve: null argument");
    const kryst_params_t* p = io.params;
    kryst_csr_t a = io.a; kryst_ctx_t ctx = a->ctx; const int64_t n = a->nrows, nt = ntiles_of(n);
    KR_ARG(bv->ctx == ctx && xv->ctx == ctx, "solve: context mismatch");
    KR_ARG(a->nrows == a->xlen && bv->n == n && xv->n == n, "solve: size mismatch");
    KR_ARG(p->restart >= 1 && p->restart <= 4096, "gmres: restart out of range");
    a->halo_started_for = nullptr;            // (csr.h: an early halo start belongs to the CG / PCG solve that made it)
    KR_ARG(p->max_iters >= 0, "solve: max_iters < 0");
    KR_HIP(hipSetDevice(ctx->device));
    kryst_pc_s pcl; kryst_pc_t pc = nullptr;
    if (io.pc) { KR_ARG(io.pc->ctx == ctx, "solve: preconditioner context"); pcl = *io.pc; if (pcl.n < 0) pcl.n = n; pc = &pcl; }
    const int side = pc ? p->precond_side : 0;        // `match (self.preconditioning, pc)`: anything else takes the `_` arm
    // side 3 -- a LABELLED EXTENSION, not in the reference: textbook left preconditioning (Arnoldi on M^-1 A from M^-1 r0 / ||M^-1 r0||,
    // Gram-Schmidt against V, in-cycle test on the preconditioned residual, cycle-end test on the true one; oracle: kro_gmres side 3).
    // The reference's own Left arm (side 1) orthogonalises against an un-normalised Z[0] and stagnates on BASELINE config 3.
    KR_ARG(side >= 0 && side <= 3, "gmres: precond_side");
    const int R = p->restart;
    Workspace ws(ctx, n);
    const int64_t n_outer = (p->max_iters + R - 1) / R;                                            // :231
    KR_TRY(ws.init(n_outer * R + 2));
    KR_TRY(ws.reserve(5 + (R + 1) + (side == 1 ? 1 : side == 2 ? R + 1 : 0)));
    // small device arrays: H, g, cs, sn, y, state, gate, pointer table
    const size_t nsmall = (size_t)(R + 1) * R + (R + 1) + 3 * (size_t)R + 64;
    double* d_small = nullptr;
    KR_HIP(hipMalloc(&d_small, sizeof(double) * nsmall + sizeof(double*) * (size_t)(R + 1)));
    ws.vecs.push_back(d_small);
    KR_HIP(hipMemsetAsync(d_small, 0, sizeof(double) * nsmall, ctx->s_main));
    GmPtrs P;
    P.h = d_small; P.g = P.h + (size_t)(R + 1) * R; P.cs = P.g + (R + 1); P.sn = P.cs + R; P.y = P.sn + R;
    P.gs = reinterpret_cast<GmState*>(P.y + R); P.restart = R;
    int* d_gate = reinterpret_cast<int*>(P.y + R + 16);
    double** d_uptr = reinterpret_cast<double**>(d_small + nsmall);
    double *xk, *r0, *w, *z, *tmp;
    KR_TRY(ws.vec(&xk)); KR_TRY(ws.vec(&r0)); KR_TRY(ws.vec(&w)); KR_TRY(ws.vec(&z)); KR_TRY(ws.vec(&tmp));
    std::vector<double*> V((size_t)R + 1), Z((size_t)R + 1, nullptr);
    for (auto& v : V) KR_TRY(ws.vec(&v));
    if (side == 1) { KR_TRY(ws.vec(&Z[0])); for (int k = 1; k <= R; ++k) Z[k] = V[k]; }            // :305-306 Z[j+1] is V[j+1]
    if (side == 2) for (auto& v : Z) KR_TRY(ws.vec(&v));
    {
        const std::vector<double*>& U = (side == 2) ? Z : V;                                      // :362-386
        KR_HIP(hipMemcpyAsync(d_uptr, U.data(), sizeof(double*) * (size_t)(R + 1), hipMemcpyHostToDevice, ctx->s_main));
        KR_HIP(hipStreamSynchronize(ctx->s_main));
    }
    const LogicCtx lc = ws.lctx(p, io.monitor != nullptr);
    LiveMonitor mon; mon.io = &io; mon.ws = &ws; mon.first = 1;
    const DevState* st = ws.st; const GmState* gs = P.gs;
    const int* done = &ws.st->done;
    int32_t rc = KRYST_OK;

    KR_HIP(hipMemcpyAsync(xk, xv->d, padded_bytes(n), hipMemcpyDeviceToDevice, ctx->s_main));      // :219
    KR_TRY(residual_dot(a, bv->d, xk, r0, tmp, nullptr));                                         // :221-227
    KR_TRY((reduce_then<1>(ctx, nt, ws.red, GmInitLogic{lc, P, side})));

    for (int64_t outer = 0; outer < n_outer; ++outer) {                                           // :234
        // ---- cycle start
        if (side == 3) {                                                                          // extension: v0 = M^-1 r0 / ||M^-1 r0||
            rc = pc_apply_dev(pc, r0, z, done); if (rc) return rc;
            KR_TRY(launch_ew(ctx, DotOneOp{z, z}, n, done));
            KR_TRY((reduce_then<1>(ctx, nt, ws.red, GmCycleLogic{lc, P, 2})));
            KR_TRY(launch_ew(ctx, DivOp{&P.gs->r0_norm, z, V[0]}, n, done));
        } else if (side == 2) {                                                                   // :248-260
            rc = pc_apply_dev(pc, r0, z, done); if (rc) return rc;
            KR_TRY(launch_ew(ctx, DotOneOp{z, z}, n, done));
            KR_TRY((reduce_then<1>(ctx, nt, ws.red, GmCycleLogic{lc, P, 1})));
            KR_TRY(launch_ew(ctx, DivOp{&P.gs->r0_norm, z, V[0]}, n, done));
            rc = pc_apply_dev(pc, V[0], Z[0], done); if (rc) return rc;
        } else {
            KR_TRY(logic_only(ctx, ws.red, GmCycleLogic{lc, P, 0}));
            KR_TRY(launch_ew(ctx, DivOp{&P.gs->r0_norm, r0, V[0]}, n, done));                     // :242 / :263
            if (side == 1) { rc = pc_apply_dev(pc, V[0], Z[0], done); if (rc) return rc; }       // :244-246
        }
        // ---- Arnoldi loop (:276-355); everything is gated on done || cyc_stop
        for (int j = 0; j < R; ++j) {
            KR_TRY(logic_only(ctx, ws.red, GmStepBeginLogic{lc, P}));
            hipLaunchKernelGGL(gate_kernel, dim3(1), dim3(1), 0, ctx->s_main, st, gs, d_gate);
            KR_HIP(hipGetLastError());
            double* zz; const std::vector<double*>& B = (side == 1) ? Z : V;
            if (side == 1) {                                                                      // :281-284
                KR_TRY(launch_spmv(a, V[j], w, 0, nullptr, d_gate));
                rc = pc_apply_dev(pc, w, z, d_gate); if (rc) return rc;
                zz = z;
                KR_TRY(launch_iter(ctx, DotOneOp{zz, B[0]}, n, st, gs));
            } else if (side == 3) {                                                               // extension: z = M^-1 A v_j against V
                KR_TRY(launch_spmv(a, V[j], w, 0, nullptr, d_gate));
                rc = pc_apply_dev(pc, w, z, d_gate); if (rc) return rc;
                zz = z;
                KR_TRY(launch_iter(ctx, DotOneOp{zz, B[0]}, n, st, gs));
            } else if (side == 2) {                                                               // :311-317
                rc = pc_apply_dev(pc, V[j], w, d_gate); if (rc) return rc;
                KR_TRY(launch_spmv(a, w, z, 1, V[0], d_gate));                                    // + (w2, V[0])
                zz = z;
            } else {                                                                              // arnoldi :79-81
                KR_TRY(launch_spmv(a, V[j], w, 1, V[0], d_gate));                                 // + (w, V[0])
                zz = w;
            }
            // double modified Gram-Schmidt (:83-96 / :286-298 / :318-330): 2(j+1) fused links
            for (int sweep = 0; sweep < 2; ++sweep)
                for (int i = 0; i <= j; ++i) {
                    KR_TRY((reduce_then<1>(ctx, nt, ws.red, GmHLogic{lc, P, i, j, sweep})));
                    const bool last = (sweep == 1 && i == j);
                    const double* nxt = last ? nullptr : (i < j ? B[i + 1] : B[0]);                    // last link: ||z||^2 (:97)
                    if (keep_in_cache(n)) KR_TRY(launch_iter(ctx, MgsLinkOp<true>{&P.gs->hcur, B[i], nxt, zz}, n, st, gs));
                    else KR_TRY(launch_iter(ctx, MgsLinkOp<false>{&P.gs->hcur, B[i], nxt, zz}, n, st, gs));
                }
            KR_TRY((reduce_then<1>(ctx, nt, ws.red, GmNormLogic{lc, P, j})));
            // v_{j+1} = z / h[j+1][j] (:102-103 / :304-306 / :336-341); skipped once the cycle is left
            KR_TRY(launch_iter(ctx, DivOp{&P.gs->hj1, zz, V[j + 1]}, n, st, gs));
            if (side == 2) {
                hipLaunchKernelGGL(gate_kernel, dim3(1), dim3(1), 0, ctx->s_main, st, gs, d_gate);
                KR_HIP(hipGetLastError());
                rc = pc_apply_dev(pc, V[j + 1], Z[j + 1], d_gate); if (rc) return rc;
            }
        }
        // ---- cycle end (:357-398)
        KR_TRY(logic_only(ctx, ws.red, GmBackLogic{lc, P}));
        KR_TRY(launch_ew(ctx, GmUpdateOp{gs, P.y, d_uptr, xk}, n, done));
        KR_TRY(residual_dot(a, bv->d, xk, r0, tmp, done));
        KR_TRY((reduce_then<1>(ctx, nt, ws.red, GmCycleEndLogic{lc, P})));
        // one host sync per restart cycle (a cycle is tens of ms of device work)
        KR_HIP(hipStreamSynchronize(ctx->s_main));
        if (ctx->nranks > 1) KR_HIP(hipStreamSynchronize(ctx->s_comm));
        mon.poll();                                                                               // live monitor: once per restart cycle
        if (ctx->h_prog->done) break;
    }
    KR_HIP(hipStreamSynchronize(ctx->s_main));
    const int32_t status = finish_solve(ws, io);
    if (status == KRYST_OK)
        KR_HIP(hipMemcpyAsync(xv->d, xk, padded_bytes(n), hipMemcpyDeviceToDevice, ctx->s_main));  // :400
    KR_HIP(hipStreamSynchronize(ctx->s_main));
    if (io.monitor) {
        DevState h;
        if (read_state(ws, &h) == hipSuccess) mon.upto(h.hist_len);
    }
    return status;
}

}  // namespace kr
