// Device-resident flexible GMRES: FgmresSolver::solve_flex (src/solver/fgmres.rs:114-340), restated literally:
// classical Gram-Schmidt by default (all h[i][j] = (w, v_i) from the SAME w, then the sequential subtractions,
// :218-236), the optional refinement sweep of OrthogMethod::Modified that does not correct h (:239-247), the
// happy-breakdown test against haptol*|s[j]| (:253), Givens/denominator == 0 -> (1,0) (:271-275), the convergence
// test against the ROTATED s[0] (:293), back-substitution without a pivot guard (:307-314), the ABSOLUTE tolerance
// on the true residual (:324) and the stats quirk final_residual = initial ||r|| (:171,:339).
//
// Classical Gram-Schmidt is what makes this solver a good fit for the GPU: the j+1 inner products of one Arnoldi
// step are independent, so they are taken 8 at a time in ONE pass over w (9n words for 8 dots instead of 16n), and the
// j+1 subtractions 8 at a time in one read-modify-write of w (10n words for 8 axpys instead of 24n); the last
// batch also produces ||w||^2.  Per basis vector the step moves 2.4n words against the reference's 5n.
#include "solver_common.h"

namespace kr {

struct FgState {                     // device
    int cyc_stop;                    // `break 'arnoldi` taken in this cycle
    int happy;                       // of the current step
    int converged;                   // the cycle-local `converged` (:205)
    int apply;                       // Modified refinement: |corr| > 1e-10 (:242)
    long long total_iters;
    long long k;                     // arnoldi_steps
    double hj1, corr, beta, beta0;
};
struct FgPtrs { FgState* fs; double* h; double* cs; double* sn; double* s; double* hcol; double* y; int ld; int restart; };

// ---- vector ops
template <int NB, bool KEEP>
struct MultiDotOp {                  // partial (w, v_k), k = 0..NB-1  (:220-222 / :231-233)
    static constexpr int NQ = NB; static constexpr const char* TAG = "MultiDot";
    const double* w; const double* v[NB];
    __device__ __forceinline__ void pair(int64_t i, bool in0, bool in1, double (&acc)[NB]) const {
        const d2 ww = ld2_sel<KEEP>(w, i);     // w is read by every kernel of the sweep: cacheable when it fits; the basis streams past it
#pragma unroll
        for (int k = 0; k < NB; ++k) {
            const d2 vv = ld2(v[k], i);
            if (in0) acc[k] = acc[k] + ww.a * vv.a;
            if (in1) acc[k] = acc[k] + ww.b * vv.b;
        }
    }
};
template <int NB, bool KEEP>
struct MultiAxpyOp {                 // w = w - h_k v_k, k ascending (:223-228 / :234-236); partial (w,w) or (w,next) of the result
    static constexpr int NQ = 1; static constexpr const char* TAG = "MultiAxpy"; static constexpr int BPC = 3;   // FGMRES(30) classical GS 256^3: 2 -> 3 workgroups per CU +1.4 %
    const double* h; const double* v[NB]; const double* next; double* w;
    __device__ __forceinline__ void pair(int64_t i, bool in0, bool in1, double (&acc)[1]) const {
        d2 ww = ld2_sel<KEEP>(w, i);
#pragma unroll
        for (int k = 0; k < NB; ++k) {
            const double hk = h[k];
            const d2 vv = ld2(v[k], i);
            ww.a = ww.a - hk * vv.a; ww.b = ww.b - hk * vv.b;
        }
        st2_sel<KEEP>(w, i, ww.a, ww.b);
        d2 nx = ww;
        if (next) nx = ld2_sel<KEEP>(next, i);
        if (in0) acc[0] = acc[0] + ww.a * nx.a;
        if (in1) acc[0] = acc[0] + ww.b * nx.b;
    }
};
template <bool KEEP>
struct RefineLinkOp {                // if |corr| > 1e-10: w = w - corr v_i (:242-246); partial (w, next) or (w, w)
    static constexpr int NQ = 1; static constexpr const char* TAG = "RefineLink";
    static constexpr int BPC = 6;    // round 5 (tools/fgmres_only.py, FGMRES(30) with Orthog::Modified): 2 / 3 / 4 / 5 / 6 / 8 workgroups per CU = 48 / 55 / 60 / 64 / 66 / 66 it/s at 512^3, 400 / 460 / 489 / - / 526 at 256^3
    const FgState* fs; const double* vi; const double* next; double* w;
    __device__ __forceinline__ void pair(int64_t i, bool in0, bool in1, double (&acc)[1]) const {
        d2 ww = ld2_sel<KEEP>(w, i);
        if (fs->apply) {
            const double c = fs->corr;
            const d2 vv = ld2(vi, i);
            ww.a = ww.a - c * vv.a; ww.b = ww.b - c * vv.b;
            st2_sel<KEEP>(w, i, ww.a, ww.b);
        }
        d2 nx = ww;
        if (next) nx = ld2_sel<KEEP>(next, i);
        if (in0) acc[0] = acc[0] + ww.a * nx.a;
        if (in1) acc[0] = acc[0] + ww.b * nx.b;
    }
};
struct NextBasisOp {                 // v_{j+1} = w / h[j+1][j], or zeros on happy breakdown (:255-261)
    static constexpr int NQ = 0; static constexpr const char* TAG = "NextBasis";
    const FgState* fs; const double* w; double* out;
    __device__ __forceinline__ void pair(int64_t i, bool, bool, double (&)[1]) const {
        if (fs->happy) { st2(out, i, 0.0, 0.0); return; }
        const double d = fs->hj1;
        const d2 a = ld2(w, i);
        st2(out, i, a.a / d, a.b / d);
    }
};
struct ScaleByOp {                   // v_0 = r / beta (:167-169, :332-334)
    static constexpr int NQ = 0; static constexpr const char* TAG = "ScaleBy";
    const double* s; const double* in; double* out;
    __device__ __forceinline__ void pair(int64_t i, bool, bool, double (&)[1]) const {
        const double d = *s;
        const d2 a = ld2(in, i);
        st2(out, i, a.a / d, a.b / d);
    }
};
struct FgUpdateOp {                  // build_solution (:344-356): x += y[i] z_i, i ascending per element
    static constexpr int NQ = 0; static constexpr const char* TAG = "FgUpdate";
    const FgState* fs; const double* y; double* const* z; double* x;
    __device__ __forceinline__ void pair(int64_t i, bool, bool, double (&)[1]) const {
        const int m = (int)fs->k;
        d2 xx = ld2(x, i);
        for (int j0 = 0; j0 < m; j0 += 8) {                  // 8 basis vectors in flight; the sum keeps its ascending order
            d2 uu[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) uu[k] = ld2(z[min(j0 + k, m - 1)], i);
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (j0 + k < m) { const double yj = y[j0 + k]; xx.a = xx.a + yj * uu[k].a; xx.b = xx.b + yj * uu[k].b; }
        }
        st2(x, i, xx.a, xx.b);
    }
};

// kernels of the Arnoldi loop are gated on done || cyc_stop
struct GateFgCycle {
    const DevState* st; const FgState* fs;
    __device__ __forceinline__ bool skip() const { return st->done || fs->cyc_stop; }
};
template <class Op>
static int32_t fg_launch(kryst_ctx_t ctx, const Op& op, int64_t n, const DevState* st, const FgState* fs, int bpc = 0) {
    static const int dflt = [] { const char* e = getenv("KRYST_FG_BLOCKS_PER_CU"); return e ? atoi(e) : 2; }();
    return launch_ew_gated(ctx, op, n, GateFgCycle{st, fs}, bpc > 0 ? bpc : std::max(dflt, ew_bpc<Op>::value));
}
__global__ void fg_gate_kernel(const DevState* st, const FgState* fs, int* gate) { *gate = (st->done || fs->cyc_stop) ? 1 : 0; }

// ---- logic
#define HH(i, k) P.h[(size_t)(i) * P.ld + (k)]
struct FgInitLogic {                 // :140-143, :166, :171 ; red0 = (r,r)
    static constexpr bool RUN_WHEN_DONE = false;
    LogicCtx c; FgPtrs P;
    __device__ void run(const double* red) const {
        DevState* st = c.st; FgState* fs = P.fs;
        const double beta = dsqrt(red[0]);
        fs->beta = beta; fs->beta0 = beta; fs->total_iters = 0; fs->cyc_stop = 0; fs->converged = 0; fs->k = 0;
        st->iterations = 0; st->final_residual = beta; st->converged = 0;
        if (beta == 0.0) { st->converged = 1; st->final_residual = 0.0; c.finish(KRYST_OK); return; }   // :141-143
        P.s[0] = beta;                                                  // :166
        if (c.max_iters <= 0) c.finish(KRYST_OK);                       // `while total_iters < max_iters` never entered
    }
};
struct FgCycleLogic {                // :203-206
    static constexpr bool RUN_WHEN_DONE = false;
    LogicCtx c; FgPtrs P; int m;
    __device__ void run(const double*) const { P.fs->cyc_stop = 0; P.fs->converged = 0; P.fs->k = m; P.fs->happy = 0; }
};
template <int NB>
struct FgHcolLogic {                 // h_col[i0 .. i0+cnt) = the folded dots
    static constexpr bool RUN_WHEN_DONE = false;
    LogicCtx c; FgPtrs P; int i0, cnt;
    __device__ void run(const double* red) const {
        if (P.fs->cyc_stop) return;
        for (int k = 0; k < cnt; ++k) P.hcol[i0 + k] = red[k];
    }
};
struct FgCorrLogic {                 // refinement: corr = (w, v_i), applied only if |corr| > 1e-10 (:241-242)
    static constexpr bool RUN_WHEN_DONE = false;
    LogicCtx c; FgPtrs P;
    __device__ void run(const double* red) const {
        if (P.fs->cyc_stop) return;
        P.fs->corr = red[0]; P.fs->apply = (fabs(red[0]) > 1e-10) ? 1 : 0;
    }
};
struct FgNormLogic {                 // :250-301 ; red0 = (w,w)
    static constexpr bool RUN_WHEN_DONE = false;
    LogicCtx c; FgPtrs P; int j; double haptol;
    __device__ void run(const double* red) const {
        FgState* fs = P.fs;
        if (fs->cyc_stop) return;
        const double hj1 = dsqrt(red[0]);
        HH(j + 1, j) = hj1; fs->hj1 = hj1;                              // :250
        for (int i = 0; i <= j; ++i) HH(i, j) = P.hcol[i];              // :251
        const double hapbnd = haptol * fabs(P.s[j]);                    // :253
        fs->happy = (fabs(hj1) < hapbnd) ? 1 : 0;                       // :254
        for (int i = 0; i < j; ++i) {                                   // :263-267
            const double temp = P.cs[i] * HH(i, j) + P.sn[i] * HH(i + 1, j);
            HH(i + 1, j) = -P.sn[i] * HH(i, j) + P.cs[i] * HH(i + 1, j);
            HH(i, j) = temp;
        }
        const double h1 = HH(j, j), h2 = HH(j + 1, j);                  // :269-278
        const double denom = dsqrt(h1 * h1 + h2 * h2);
        double cc, ss;
        if (denom == 0.0) { cc = 1.0; ss = 0.0; } else { cc = h1 / denom; ss = h2 / denom; }
        P.cs[j] = cc; P.sn[j] = ss;
        const double temp = cc * P.s[j] + ss * P.s[j + 1];              // :281-283
        P.s[j + 1] = -ss * P.s[j] + cc * P.s[j + 1];
        P.s[j] = temp;
        HH(j, j) = cc * HH(j, j) + ss * HH(j + 1, j);                   // :284-285
        HH(j + 1, j) = 0.0;
        const double res_norm = fabs(P.s[j + 1]);                       // :286
        fs->total_iters = fs->total_iters + 1;                          // :287
        c.push(res_norm);                                               // :289-292
        const bool stop = c.check(res_norm, P.s[0], fs->total_iters);   // :293 (s[0] already rotated)
        if (stop) { fs->k = j + 1; fs->converged = 1; fs->cyc_stop = 1; }   // :295-301
    }
};
struct FgBackLogic {                 // :304-314 (no pivot guard)
    static constexpr bool RUN_WHEN_DONE = false;
    LogicCtx c; FgPtrs P;
    __device__ void run(const double*) const {
        const int k = (int)P.fs->k;
        for (int i = k - 1; i >= 0; --i) {
            double sum = P.s[i];
            for (int l = i + 1; l < k; ++l) sum = sum - HH(i, l) * P.y[l];
            P.y[i] = sum / HH(i, i);
        }
    }
};
struct FgCycleEndLogic {             // :323-337 then :339-340 ; red0 = (r_new, r_new)
    static constexpr bool RUN_WHEN_DONE = false;
    LogicCtx c; FgPtrs P; int last_cycle;
    __device__ void run(const double* red) const {
        DevState* st = c.st; FgState* fs = P.fs;
        const double res_true = dsqrt(red[0]);
        st->iter = fs->total_iters;
        bool leave = false;
        if (res_true < c.tol || fs->converged) { st->converged = 1; leave = true; }   // :324-329 (absolute tolerance)
        else {
            fs->beta = res_true;                                        // :331
            for (int q = 0; q <= P.restart; ++q) P.s[q] = 0.0;          // :335-337
            P.s[0] = res_true;
            if (last_cycle) leave = true;                               // `while total_iters < max_iters` ends
        }
        if (leave) {
            st->final_residual = fs->beta0;                             // :339: the outer `res_norm` is the initial ||r||
            st->iterations = fs->total_iters;                           // :340
            c.finish(KRYST_OK);
        }
    }
};
#undef HH

template <class L>
static int32_t fg_logic_only(kryst_ctx_t ctx, const double* red, const L& l) {
    hipLaunchKernelGGL((logic_kernel<L>), dim3(1), dim3(64), 0, ctx->s_main, red, l);
    KR_HIP(hipGetLastError());
    return KRYST_OK;
}

// one batch of up to 8 dots + its fold; slots past cnt alias the first vector and are ignored by the logic
template <int NB>
static int32_t dot_batch(kryst_ctx_t ctx, int64_t n, int64_t nt, double* red, const LogicCtx& lc, const FgPtrs& P,
                         const DevState* st, const double* w, double* const* v, int i0, int cnt) {
    static const int bpc = [] { const char* e = getenv("KRYST_DOT_BLOCKS_PER_CU"); return e ? atoi(e) : 4; }();   // read-only reductions want more waves in flight than the mixed streams
    auto go = [&](auto op) -> int32_t {
        op.w = w;
        for (int k = 0; k < NB; ++k) op.v[k] = v[i0 + (k < cnt ? k : 0)];
        return fg_launch(ctx, op, n, st, P.fs, bpc);
    };
    if (keep_in_cache(n)) KR_TRY(go(MultiDotOp<NB, true>{})); else KR_TRY(go(MultiDotOp<NB, false>{}));
    return reduce_then<NB>(ctx, nt, red, FgHcolLogic<NB>{lc, P, i0, cnt});
}
template <int NB>
static int32_t axpy_batch(kryst_ctx_t ctx, int64_t n, const FgPtrs& P, const DevState* st, double* w, double* const* v,
                          int i0, const double* next) {
    auto go = [&](auto op) -> int32_t {
        op.h = P.hcol + i0; op.next = next; op.w = w;
        for (int k = 0; k < NB; ++k) op.v[k] = v[i0 + k];
        return fg_launch(ctx, op, n, st, P.fs);
    };
    return keep_in_cache(n) ? go(MultiAxpyOp<NB, true>{}) : go(MultiAxpyOp<NB, false>{});
}

int32_t fgmres_solve(kryst_vec_t bv, kryst_vec_t xv, const SolveIO& io, int32_t orthog, double haptol, int32_t preallocate) {
    const EnvFreeze knobs;                    // the tuning knobs are read once per solve, not per launch
    KR_ARG(io.a && io.params && bv && xv, "solve: null argument");
    const kryst_params_t* p = io.params;
    kryst_csr_t a = io.a; kryst_ctx_t ctx = a->ctx; const int64_t n = a->nrows, nt = ntiles_of(n);
    KR_ARG(bv->ctx == ctx && xv->ctx == ctx, "solve: context mismatch");
    KR_ARG(a->nrows == a->xlen && bv->n == n && xv->n == n, "solve: size mismatch");
    KR_ARG(p->restart >= 1 && p->restart <= 4096, "fgmres: restart out of range (restart = 0 never terminates in the reference)");
    a->halo_started_for = nullptr;            // (csr.h: an early halo start belongs to the CG / PCG solve that made it)
    KR_ARG(p->max_iters >= 0, "solve: max_iters < 0");
    KR_ARG(orthog == 0 || orthog == 1, "fgmres: orthog (0 Classical, 1 Modified)");
    KR_HIP(hipSetDevice(ctx->device));
    kryst_pc_s pcl; kryst_pc_t pc = nullptr;
    if (io.pc && io.pc->kind != KR_PC_IDENTITY) { KR_ARG(io.pc->ctx == ctx, "solve: preconditioner context"); pcl = *io.pc; if (pcl.n < 0) pcl.n = n; pc = &pcl; }
    const int64_t max_iters = p->max_iters;
    const int R = (int)std::min<int64_t>(p->restart, std::max<int64_t>(max_iters, 1));    // no cycle is ever longer than this
    Workspace ws(ctx, n);
    KR_TRY(ws.init(max_iters + (int64_t)p->restart + 2));
    KR_TRY(ws.reserve(4 + (R + 1) + (pc ? R : 0)));
    // small device arrays: H (R+1 x R), cs, sn, s, hcol, y, state, gate, pointer table
    const size_t nsmall = (size_t)(R + 1) * R + 2 * (size_t)R + (size_t)p->restart + 2 + 2 * (size_t)(R + 8) + 64;
    double* d_small = nullptr;
    KR_HIP(hipMalloc(&d_small, sizeof(double) * nsmall + sizeof(double*) * (size_t)(R + 1)));
    ws.vecs.push_back(d_small);
    KR_HIP(hipMemsetAsync(d_small, 0, sizeof(double) * nsmall, ctx->s_main));
    FgPtrs P;
    P.h = d_small; P.cs = P.h + (size_t)(R + 1) * R; P.sn = P.cs + R; P.s = P.sn + R; P.hcol = P.s + p->restart + 2;
    P.y = P.hcol + R + 8; P.fs = reinterpret_cast<FgState*>(P.y + R + 8); P.ld = R; P.restart = p->restart;
    int* d_gate = reinterpret_cast<int*>(P.y + R + 8 + 32);
    double** d_zptr = reinterpret_cast<double**>(d_small + nsmall);
    double *xk, *r, *w, *tmp;
    KR_TRY(ws.vec(&xk)); KR_TRY(ws.vec(&r)); KR_TRY(ws.vec(&w)); KR_TRY(ws.vec(&tmp));
    std::vector<double*> V((size_t)R + 1), Z((size_t)R + 1, nullptr);
    for (auto& v : V) KR_TRY(ws.vec(&v));
    // z_j = v_j.clone() then pc.apply (:209-212): without a preconditioner z_j IS v_j, so the copy is not materialised
    if (pc) { for (int k = 0; k < R; ++k) KR_TRY(ws.vec(&Z[k])); } else Z = V;
    KR_HIP(hipMemcpyAsync(d_zptr, Z.data(), sizeof(double*) * (size_t)(R + 1), hipMemcpyHostToDevice, ctx->s_main));
    KR_HIP(hipStreamSynchronize(ctx->s_main));
    const LogicCtx lc = ws.lctx(p, io.monitor != nullptr);
    LiveMonitor mon; mon.io = &io; mon.ws = &ws; mon.first = 1;
    const DevState* st = ws.st; const FgState* fs = P.fs;
    const int* done = &ws.st->done;
    int32_t rc = KRYST_OK;

    KR_HIP(hipMemcpyAsync(xk, xv->d, padded_bytes(n), hipMemcpyDeviceToDevice, ctx->s_main));
    KR_TRY(residual_dot(a, bv->d, xk, r, tmp, nullptr));                                          // :133-140
    KR_TRY((reduce_then<1>(ctx, nt, ws.red, FgInitLogic{lc, P})));

    int64_t total = 0;                       // total_iters at the start of the cycle (a cycle that stops early ends the solve)
    while (total < max_iters) {                                                                   // :175
        const int m = (int)(preallocate ? std::min<int64_t>(max_iters, p->restart)
                                        : std::min<int64_t>(p->restart, max_iters - total));      // :203
        KR_TRY(fg_logic_only(ctx, ws.red, FgCycleLogic{lc, P, m}));
        KR_TRY(launch_ew(ctx, ScaleByOp{&P.fs->beta, r, V[0]}, n, done));                         // :167-169 / :332-334
        for (int j = 0; j < m; ++j) {                                                             // :207
            hipLaunchKernelGGL(fg_gate_kernel, dim3(1), dim3(1), 0, ctx->s_main, st, fs, d_gate);
            KR_HIP(hipGetLastError());
            if (pc) { rc = pc_apply_dev(pc, V[j], Z[j], d_gate); if (rc) return rc; }             // :209-212
            KR_TRY(launch_spmv(a, Z[j], w, 0, nullptr, d_gate));                                  // :214-215
            // all h_col[i] = (w, v_i) from the unmodified w (:220-222 / :231-233)
            for (int i0 = 0; i0 <= j; i0 += 8) {
                const int cnt = std::min(8, j + 1 - i0);
                if (cnt > 4) KR_TRY(dot_batch<8>(ctx, n, nt, ws.red, lc, P, st, w, V.data(), i0, cnt));
                else if (cnt > 2) KR_TRY(dot_batch<4>(ctx, n, nt, ws.red, lc, P, st, w, V.data(), i0, cnt));
                else if (cnt > 1) KR_TRY(dot_batch<2>(ctx, n, nt, ws.red, lc, P, st, w, V.data(), i0, cnt));
                else KR_TRY(dot_batch<1>(ctx, n, nt, ws.red, lc, P, st, w, V.data(), i0, cnt));
            }
            // w -= h_col[i] v_i, i ascending (:223-228 / :234-236); the last batch carries (w,w) or the first refinement dot
            for (int i0 = 0; i0 <= j; ) {
                const int left = j + 1 - i0;
                const int nb = left >= 8 ? 8 : left >= 4 ? 4 : left >= 2 ? 2 : 1;
                const bool last = (i0 + nb == j + 1);
                const double* next = (last && orthog == 1) ? V[0] : nullptr;
                if (nb == 8) KR_TRY(axpy_batch<8>(ctx, n, P, st, w, V.data(), i0, next));
                else if (nb == 4) KR_TRY(axpy_batch<4>(ctx, n, P, st, w, V.data(), i0, next));
                else if (nb == 2) KR_TRY(axpy_batch<2>(ctx, n, P, st, w, V.data(), i0, next));
                else KR_TRY(axpy_batch<1>(ctx, n, P, st, w, V.data(), i0, next));
                i0 += nb;
            }
            if (orthog == 1)                                                                      // :239-247
                for (int i = 0; i <= j; ++i) {
                    KR_TRY((reduce_then<1>(ctx, nt, ws.red, FgCorrLogic{lc, P})));
                    if (keep_in_cache(n)) KR_TRY(fg_launch(ctx, RefineLinkOp<true>{fs, V[i], i < j ? V[i + 1] : nullptr, w}, n, st, fs));
                    else KR_TRY(fg_launch(ctx, RefineLinkOp<false>{fs, V[i], i < j ? V[i + 1] : nullptr, w}, n, st, fs));
                }
            KR_TRY((reduce_then<1>(ctx, nt, ws.red, FgNormLogic{lc, P, j, haptol})));
            KR_TRY(fg_launch(ctx, NextBasisOp{fs, w, V[j + 1]}, n, st, fs, 4));                   // :255-261 (fp64 divisions: 4 workgroups per CU)
        }
        // ---- cycle end (:303-337)
        total += m;
        KR_TRY(fg_logic_only(ctx, ws.red, FgBackLogic{lc, P}));
        KR_TRY(launch_ew(ctx, FgUpdateOp{fs, P.y, d_zptr, xk}, n, done));
        KR_TRY(residual_dot(a, bv->d, xk, r, tmp, done));
        KR_TRY((reduce_then<1>(ctx, nt, ws.red, FgCycleEndLogic{lc, P, total >= max_iters ? 1 : 0})));
        KR_HIP(hipStreamSynchronize(ctx->s_main));                  // one host sync per restart cycle
        if (ctx->nranks > 1) KR_HIP(hipStreamSynchronize(ctx->s_comm));
        mon.poll();                                                                               // live monitor: once per restart cycle
        if (ctx->h_prog->done) break;
    }
    KR_HIP(hipStreamSynchronize(ctx->s_main));
    const int32_t status = finish_solve(ws, io);
    if (status == KRYST_OK) KR_HIP(hipMemcpyAsync(xv->d, xk, padded_bytes(n), hipMemcpyDeviceToDevice, ctx->s_main));
    KR_HIP(hipStreamSynchronize(ctx->s_main));
    if (io.monitor) {
        DevState h;
        if (read_state(ws, &h) == hipSuccess) mon.upto(h.hist_len);
    }
    return status;
}

}  // namespace kr
