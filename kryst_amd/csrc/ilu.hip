// ILU(0)-family preconditioners with a level-scheduled triangular solve on the device.
//
//   mode KRYST_ILU_KRYST_COMPAT  Ilu0 exactly as written (src/preconditioner/ilu.rs:59-122): the net effect of its setup is
//                                L = I + tril(A,-1) D^-1 (l_ij = a_ij / a_jj), U = triu(A); apply never divides by u_ii.
//   mode KRYST_ILU_ILUP0         Ilup::new(0) exactly as written (src/preconditioner/ilup.rs:77-167): no elimination happens
//                                at fill 0 (new_level = 1 > fill), same L, U = triu(A), apply divides by the stored diagonal.
//   mode KRYST_ILU_TRUE_ILU0     extension: textbook IKJ ILU(0) on A's pattern (Saad Alg. 10.4).
//
// Setup (host, once): factor values, dependency levels of L (ascending rows) and U (descending rows), and a LEVEL-ORDERED
// copy of each factor (rows of one level contiguous, columns renumbered to level-order positions).
// Apply (device): the vectors are permuted into level order, the two triangular systems are solved -- a lane owns a row and
// subtracts its entries in stored order, exactly the reference's loop -- and the result is permuted back; the launch sequence
// is captured once into a hipGraph.  A 7-point 256^3 grid has 3N-2 = 766 dependency levels per factor, so the solve is
// LATENCY-bound.  Two forms with identical results: one kernel per level (3.4 us per level: kernel boundary + ~3 dependent
// memory round trips; KRYST_ILU_SYNCFREE=0) and, by default, ONE sync-free launch per factor in which every lane polls the
// solution entries it needs (2.05 us per dependency hop, 3.1 instead of 5.2 ms per apply; see tri_syncfree_ell_kernel).
// A band schedule (B levels per launch, each workgroup recomputing the in-band dependency closure of its rows so that
// workgroup barriers replace kernel boundaries) was built and measured: bit-identical but no faster, removed again.
// In a distributed context the factors are block-local (block-Jacobi ILU): halo columns are dropped.
#include "pc.h"
#include "ew.h"
#include <algorithm>
#include <cmath>
#include <map>

namespace kr {

static int env_i(const char* nm, int d) { const char* e = getenv(nm); return e ? atoi(e) : d; }

struct TriArgs {                    // device-resident argument block, rewritten before every apply (graph-friendly):
    const double* r; double* z; long long skip;   // one scalar load gives a level kernel everything it needs
};

#define ELLW 4
struct EllView { const int32_t* col; const double* val; const uint8_t* len; int64_t npos; };

struct TriFactor {                  // one triangular factor in level order
    int32_t* d_ptr = nullptr;       // npos+1: entry range of the row at level-position p
    int32_t* d_col = nullptr;       // column (original numbering)
    double*  d_val = nullptr;
    int32_t* d_row = nullptr;       // npos: original row id
    double*  d_diag = nullptr;      // npos: divisor (1.0 when the apply does not divide)
    // ELL copy (rows of <= ELLW kept entries, e.g. any 7-point factor): slot-major, so a lane's loads do not depend
    // on a row pointer -- one round trip less on a latency-bound kernel
    int32_t* d_ecol = nullptr; double* d_eval = nullptr; uint8_t* d_elen = nullptr; int64_t npos = 0; bool ell = false;
    bool syncfree = false;          // ELL factor solved by ONE sync-free launch (KRYST_ILU_SYNCFREE=0: one launch per level)
    std::vector<int32_t> lvl_off;   // host: position offsets per level
    int32_t* d_lvl_off = nullptr;
    void free_all() { (void)hipFree(d_ptr); (void)hipFree(d_col); (void)hipFree(d_val); (void)hipFree(d_row); (void)hipFree(d_diag); (void)hipFree(d_lvl_off);
                      (void)hipFree(d_ecol); (void)hipFree(d_eval); (void)hipFree(d_elen); }
    EllView view() const { return EllView{d_ecol, d_eval, d_elen, npos}; }
};

struct IluData {
    TriFactor L, U;
    TriArgs* d_args = nullptr;
    double* d_rL = nullptr; double* d_y = nullptr; double* d_yU = nullptr; double* d_zU = nullptr;   // level-permuted work vectors
    int32_t* d_mapLU = nullptr;     // L-position of the row at U-position q
    hipGraph_t graph = nullptr; hipGraphExec_t exec = nullptr;
    int64_t n = 0;
};

// forward:  y[i] = r[i] - sum l_ij y[j]            (ilu.rs:107-113, ilup.rs:143-149)
// backward: z[i] = (y[i] - sum u_ij z[j]) / d_i     (ilu.rs:115-119 with d = 1, ilup.rs:151-165)
// All level kernels work on LEVEL-PERMUTED vectors (`in`, `out` indexed by the level-order position p; the factor's column
// indices are positions too): the rows of a level and, for banded operators, their dependencies in the previous level are
// contiguous in memory, so a level touches a few pages instead of one page per row (the scattered form spent ~5 us per
// level on address translation and uncoalesced 8-byte accesses).  perm_gather / perm_scatter convert at the ends.

// rows of ONE level: positions [p0, p1)
template <bool FORWARD>
__global__ __launch_bounds__(256) void tri_level_kernel(const TriArgs* args, const double* __restrict__ in, double* out,
                                                        const int32_t* __restrict__ ptr, const int32_t* __restrict__ col,
                                                        const double* __restrict__ val, const double* __restrict__ diag,
                                                        int32_t p0, int32_t p1) {
    if (args->skip) return;
    const int32_t p = p0 + blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= p1) return;
    double s = in[p];
    for (int32_t k = ptr[p]; k < ptr[p + 1]; ++k) s = s - val[k] * out[col[k]];    // stored order
    out[p] = FORWARD ? s : s / diag[p];                                             // diag == 1.0: exact no-op
}

__device__ __forceinline__ double ell_row(const EllView& E, int32_t p, double s, const double* z) {
    int32_t c[ELLW]; double v[ELLW], zz[ELLW];
    const int len = E.len[p];
#pragma unroll
    for (int u = 0; u < ELLW; ++u) { c[u] = E.col[u * E.npos + p]; v[u] = E.val[u * E.npos + p]; }
#pragma unroll
    for (int u = 0; u < ELLW; ++u) zz[u] = z[c[u]];                 // padding slots point at column 0 (valid, unused)
#pragma unroll
    for (int u = 0; u < ELLW; ++u) if (u < len) s = s - v[u] * zz[u];   // stored order
    return s;
}

template <bool FORWARD>
__global__ __launch_bounds__(256) void tri_level_ell_kernel(const TriArgs* args, const double* __restrict__ in, double* out, EllView E,
                                                            const double* __restrict__ diag, int32_t p0, int32_t p1) {
    if (args->skip) return;
    const int32_t p = p0 + blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= p1) return;
    const double s = ell_row(E, p, in[p], out);
    out[p] = FORWARD ? s : s / diag[p];
}

// SYNC-FREE form (ELL factors): ONE launch per factor instead of one per level.  Lane = level-order position; a lane polls
// the solution entries its row depends on until they have been written (the solution vector is pre-filled with a NaN
// sentinel, so the value is its own ready flag: one 8-byte agent-scope load per dependency and poll), then subtracts them in
// the stored order -- the same arithmetic as the level kernels -- and publishes its own entry with an agent-scope store.
// Forward progress: dependencies live at SMALLER positions (earlier levels), workgroups are dispatched in index order per
// XCD, so the lowest unfinished workgroup is always resident and depends only on finished ones.  Lanes of one wave may
// depend on each other (a wave can span several small levels), so nobody blocks: every round each unfinished lane polls
// once, ready lanes finish, and the wave leaves when all its lanes have.  A poll budget turns a logic error into NaNs
// instead of a hung GPU.
#define KR_TRI_SENTINEL 0xFFF8DEADBEEFCAFEull
template <bool FORWARD>
__global__ __launch_bounds__(256) void tri_syncfree_ell_kernel(const TriArgs* args, const double* __restrict__ in, double* out, EllView E,
                                                               const double* __restrict__ diag, int32_t npos) {
    if (args->skip) return;
    const int32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    const bool active = p < npos;
    int len = 0; int32_t c[ELLW]; double v[ELLW], zz[ELLW]; double s = 0.0, dg = 1.0;
#pragma unroll
    for (int u = 0; u < ELLW; ++u) { c[u] = 0; v[u] = 0.0; zz[u] = 0.0; }
    if (active) {
        len = E.len[p];
#pragma unroll
        for (int u = 0; u < ELLW; ++u) { c[u] = E.col[u * E.npos + p]; v[u] = E.val[u * E.npos + p]; }
        s = in[p];
        if (!FORWARD) dg = diag[p];
    }
    const unsigned want = (1u << len) - 1u;
    unsigned have = 0;
    bool done = !active;
    for (int budget = 1 << 22; budget > 0; --budget) {
        if (!done) {
#pragma unroll
            for (int u = 0; u < ELLW; ++u)
                if (u < len && !((have >> u) & 1u)) {
                    const double x = __hip_atomic_load(&out[c[u]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if ((unsigned long long)__double_as_longlong(x) != KR_TRI_SENTINEL) { zz[u] = x; have |= 1u << u; }
                }
            if (have == want) {
#pragma unroll
                for (int u = 0; u < ELLW; ++u) if (u < len) s = s - v[u] * zz[u];       // stored order
                __hip_atomic_store(&out[p], FORWARD ? s : s / dg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                done = true;
            }
        }
        if (__all(done)) return;
        __builtin_amdgcn_s_sleep(1);                        // the poll period does not matter (0..4 measured alike): the hop is two fabric trips
    }
    if (!done) __hip_atomic_store(&out[p], __longlong_as_double(0x7FF8000000000000ll), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // budget exhausted
}

// the same for factors with longer rows (CSR in level order): a lane advances through its entries as far as they are ready
template <bool FORWARD>
__global__ __launch_bounds__(256) void tri_syncfree_csr_kernel(const TriArgs* args, const double* __restrict__ in, double* out,
                                                               const int32_t* __restrict__ ptr, const int32_t* __restrict__ col,
                                                               const double* __restrict__ val, const double* __restrict__ diag, int32_t npos) {
    if (args->skip) return;
    const int32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    const bool active = p < npos;
    int32_t k = 0, kend = 0; double s = 0.0, dg = 1.0;
    if (active) { k = ptr[p]; kend = ptr[p + 1]; s = in[p]; if (!FORWARD) dg = diag[p]; }
    bool done = !active;
    for (int budget = 1 << 22; budget > 0; --budget) {
        if (!done) {
            while (k < kend) {
                const double x = __hip_atomic_load(&out[col[k]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if ((unsigned long long)__double_as_longlong(x) == KR_TRI_SENTINEL) break;
                s = s - val[k] * x;                                                        // stored order
                ++k;
            }
            if (k == kend) {
                __hip_atomic_store(&out[p], FORWARD ? s : s / dg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                done = true;
            }
        }
        if (__all(done)) return;
        __builtin_amdgcn_s_sleep(1);
    }
    if (!done) __hip_atomic_store(&out[p], __longlong_as_double(0x7FF8000000000000ll), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // budget exhausted
}

// a run of consecutive NARROW levels [l0, l1) in one workgroup (CSR fallback for factors that do not fit the ELL form)
template <bool FORWARD>
__global__ __launch_bounds__(1024) void tri_run_kernel(const TriArgs* args, const double* __restrict__ in, double* out,
                                                       const int32_t* __restrict__ ptr, const int32_t* __restrict__ col,
                                                       const double* __restrict__ val, const double* __restrict__ diag,
                                                       const int32_t* __restrict__ lvl_off, int32_t l0, int32_t l1) {
    if (args->skip) return;
    for (int32_t lv = l0; lv < l1; ++lv) {
        const int32_t p0 = lvl_off[lv], p1 = lvl_off[lv + 1];
        for (int32_t p = p0 + threadIdx.x; p < p1; p += blockDim.x) {
            double s = in[p];
            for (int32_t k = ptr[p]; k < ptr[p + 1]; ++k) s = s - val[k] * out[col[k]];
            out[p] = FORWARD ? s : s / diag[p];
        }
        __syncthreads();
    }
}

// runs in stream order before the level kernels, so it sees the solver's `done` flag as of this apply
__global__ void tri_set_args(TriArgs* a, const double* r, double* z, const int* done) { a->r = r; a->z = z; a->skip = (done && *done) ? 1 : 0; }

// dst[p] = src[map[p]]   (MODE 0: src = the caller's r;  MODE 1: plain gather;  MODE 2: scatter into the caller's z: z[map[p]] = src[p])
template <int MODE>
__global__ __launch_bounds__(256) void perm_kernel(const TriArgs* args, double* dst, const double* src, const int32_t* __restrict__ map, int64_t n,
                                                   double* fill) {
    if (args->skip) return;
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    if (MODE == 0) dst[p] = args->r[map[p]];
    else if (MODE == 1) dst[p] = src[map[p]];
    else args->z[map[p]] = src[p];
    if (fill) fill[p] = __longlong_as_double((long long)KR_TRI_SENTINEL);     // "not yet solved" marks for the sync-free kernel
}

static const int NARROW = 2048;     // levels with at most this many rows are folded into one-workgroup runs (CSR fallback)

template <bool FORWARD>
static int32_t enqueue_factor(hipStream_t s, const TriFactor& F, const TriArgs* d_args, const double* in, double* out) {
    if (F.syncfree && F.npos > 0) {
        const dim3 grid((unsigned)((F.npos + 255) / 256));
        if (F.ell) hipLaunchKernelGGL((tri_syncfree_ell_kernel<FORWARD>), grid, dim3(256), 0, s, d_args, in, out, F.view(), F.d_diag, (int32_t)F.npos);
        else hipLaunchKernelGGL((tri_syncfree_csr_kernel<FORWARD>), grid, dim3(256), 0, s, d_args, in, out, F.d_ptr, F.d_col, F.d_val, F.d_diag,
                                (int32_t)F.npos);
        KR_HIP(hipGetLastError());
        return KRYST_OK;
    }
    const int nl = (int)F.lvl_off.size() - 1;
    int lv = 0;
    while (lv < nl) {
        const int rows = F.lvl_off[lv + 1] - F.lvl_off[lv];
        if (rows <= NARROW && !F.ell) {
            int l1 = lv + 1;
            while (l1 < nl && F.lvl_off[l1 + 1] - F.lvl_off[l1] <= NARROW) ++l1;
            hipLaunchKernelGGL((tri_run_kernel<FORWARD>), dim3(1), dim3(1024), 0, s, d_args, in, out, F.d_ptr, F.d_col, F.d_val,
                               F.d_diag, F.d_lvl_off, lv, l1);
            lv = l1;
        } else {
            if (F.ell)
                hipLaunchKernelGGL((tri_level_ell_kernel<FORWARD>), dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, s, d_args, in, out,
                                   F.view(), F.d_diag, F.lvl_off[lv], F.lvl_off[lv + 1]);
            else
                hipLaunchKernelGGL((tri_level_kernel<FORWARD>), dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, s, d_args, in, out, F.d_ptr,
                                   F.d_col, F.d_val, F.d_diag, F.lvl_off[lv], F.lvl_off[lv + 1]);
            lv += 1;
        }
        KR_HIP(hipGetLastError());
    }
    return KRYST_OK;
}

// r -> rL (L level order) -> forward -> yL -> yU (U level order) -> backward -> zU -> z
static int32_t enqueue_apply(hipStream_t s, IluData* D) {
    const unsigned g = (unsigned)((D->n + 255) / 256);
    hipLaunchKernelGGL((perm_kernel<0>), dim3(g), dim3(256), 0, s, D->d_args, D->d_rL, (const double*)nullptr, D->L.d_row, D->n,
                       D->L.syncfree ? D->d_y : (double*)nullptr);
    KR_HIP(hipGetLastError());
    KR_TRY(enqueue_factor<true>(s, D->L, D->d_args, D->d_rL, D->d_y));
    hipLaunchKernelGGL((perm_kernel<1>), dim3(g), dim3(256), 0, s, D->d_args, D->d_yU, (const double*)D->d_y, D->d_mapLU, D->n,
                       D->U.syncfree ? D->d_zU : (double*)nullptr);
    KR_HIP(hipGetLastError());
    KR_TRY(enqueue_factor<false>(s, D->U, D->d_args, D->d_yU, D->d_zU));
    hipLaunchKernelGGL((perm_kernel<2>), dim3(g), dim3(256), 0, s, D->d_args, (double*)nullptr, (const double*)D->d_zU, D->U.d_row, D->n,
                       (double*)nullptr);
    KR_HIP(hipGetLastError());
    return KRYST_OK;
}

int32_t ilu_apply_dev(kryst_pc_t pc, const double* r, double* z, const int* done) {
    IluData* D = reinterpret_cast<IluData*>(pc->d_work);
    kryst_ctx_t ctx = pc->ctx;
    if (D->n == 0) return KRYST_OK;
    hipLaunchKernelGGL(tri_set_args, dim3(1), dim3(1), 0, ctx->s_main, D->d_args, r, z, done);
    KR_HIP(hipGetLastError());
    static const int use_graph = getenv("KRYST_ILU_GRAPH") ? atoi(getenv("KRYST_ILU_GRAPH")) : 1;
    if (!D->exec && use_graph) {
        // capture the launch sequence once; the graph only refers to the device argument block
        hipGraph_t g = nullptr;
        if (hipStreamBeginCapture(ctx->s_main, hipStreamCaptureModeThreadLocal) == hipSuccess) {
            int32_t rc = enqueue_apply(ctx->s_main, D);
            hipError_t e = hipStreamEndCapture(ctx->s_main, &g);
            if (rc == KRYST_OK && e == hipSuccess && g && hipGraphInstantiate(&D->exec, g, nullptr, nullptr, 0) == hipSuccess) {
                D->graph = g;
            } else {
                if (g) (void)hipGraphDestroy(g);
                D->exec = nullptr;
                (void)hipGetLastError();
            }
        }
    }
    if (D->exec) { KR_HIP(hipGraphLaunch(D->exec, ctx->s_main)); return KRYST_OK; }
    return enqueue_apply(ctx->s_main, D);                             // eager fallback (same kernels)
}

void ilu_free(kryst_pc_t pc) {
    if (pc->kind != KR_PC_ILU || !pc->d_work) return;
    IluData* D = reinterpret_cast<IluData*>(pc->d_work);
    if (D->exec) (void)hipGraphExecDestroy(D->exec);
    if (D->graph) (void)hipGraphDestroy(D->graph);
    D->L.free_all(); D->U.free_all(); (void)hipFree(D->d_args); (void)hipFree(D->d_y); (void)hipFree(D->d_rL); (void)hipFree(D->d_yU); (void)hipFree(D->d_zU); (void)hipFree(D->d_mapLU);
    delete D;
    pc->d_work = nullptr;
}

template <class T>
static int32_t up(T** dst, const std::vector<T>& v) {
    KR_HIP(hipMalloc(dst, sizeof(T) * (v.size() + 1)));
    if (!v.empty()) KR_HIP(hipMemcpy(*dst, v.data(), sizeof(T) * v.size(), hipMemcpyHostToDevice));
    return KRYST_OK;
}


// kept entries of a triangular factor, row by row in STORED order (flat CSR: a vector per row costs 2n heap blocks)
struct FlatRows {
    std::vector<int64_t> ptr; std::vector<int32_t> col; std::vector<double> val;
    int64_t len(int64_t i) const { return ptr[(size_t)i + 1] - ptr[(size_t)i]; }
};

// level order of one factor
static int32_t build_factor(int64_t n, const FlatRows& ent, const std::vector<double>& diag, bool forward, TriFactor* F,
                            std::vector<int32_t>* pos_out) {
    std::vector<int32_t> lvl((size_t)n, 0);
    int32_t nl = 0;
    auto level_of = [&](int64_t i) {
        int32_t l = 0;
        for (int64_t k = ent.ptr[i]; k < ent.ptr[i + 1]; ++k) l = std::max(l, lvl[ent.col[k]] + 1);
        lvl[i] = l; nl = std::max(nl, l + 1);
    };
    if (forward) for (int64_t i = 0; i < n; ++i) level_of(i);
    else for (int64_t i = n - 1; i >= 0; --i) level_of(i);
    F->lvl_off.assign((size_t)nl + 1, 0);
    for (int64_t i = 0; i < n; ++i) F->lvl_off[lvl[i] + 1]++;
    for (int l = 0; l < nl; ++l) F->lvl_off[l + 1] += F->lvl_off[l];
    std::vector<int32_t> cursor(F->lvl_off.begin(), F->lvl_off.end() - 1), rowid((size_t)n), ptr((size_t)n + 1, 0);
    for (int64_t i = 0; i < n; ++i) rowid[cursor[lvl[i]]++] = (int32_t)i;        // ascending row inside a level
    std::vector<int32_t> pos((size_t)n);
    for (int64_t p = 0; p < n; ++p) pos[rowid[p]] = (int32_t)p;
    if (pos_out) *pos_out = pos;
    const size_t nnz = ent.col.size();
    std::vector<int32_t> col(nnz); std::vector<double> val(nnz), dg((size_t)n);
    size_t w = 0; int64_t maxlen = 0;
    for (int64_t p = 0; p < n; ++p) {
        const int32_t i = rowid[p];
        for (int64_t k = ent.ptr[i]; k < ent.ptr[i + 1]; ++k) { col[w] = pos[ent.col[k]]; val[w] = ent.val[k]; ++w; }   // columns as level-order positions
        ptr[p + 1] = (int32_t)w;
        dg[p] = diag[i];
        maxlen = std::max(maxlen, ent.len(i));
    }
    F->npos = n;
    if (maxlen <= ELLW && n > 0) {
        std::vector<int32_t> ecol((size_t)ELLW * n, 0); std::vector<double> eval((size_t)ELLW * n, 0.0); std::vector<uint8_t> elen((size_t)n, 0);
        for (int64_t p = 0; p < n; ++p) {
            const int64_t len = ptr[p + 1] - ptr[p];
            elen[p] = (uint8_t)len;
            for (int64_t u = 0; u < len; ++u) { ecol[(size_t)u * n + p] = col[ptr[p] + u]; eval[(size_t)u * n + p] = val[ptr[p] + u]; }
        }
        KR_TRY(up(&F->d_ecol, ecol)); KR_TRY(up(&F->d_eval, eval)); KR_TRY(up(&F->d_elen, elen));
        F->ell = true;
    }
    F->syncfree = env_i("KRYST_ILU_SYNCFREE", 1) != 0;
    KR_TRY(up(&F->d_ptr, ptr)); KR_TRY(up(&F->d_col, col)); KR_TRY(up(&F->d_val, val)); KR_TRY(up(&F->d_row, rowid));
    KR_TRY(up(&F->d_diag, dg)); KR_TRY(up(&F->d_lvl_off, F->lvl_off));
    return KRYST_OK;
}

}  // namespace kr

using namespace kr;

typedef std::vector<std::vector<std::pair<int32_t, double>>> RowLists;
static FlatRows flatten(const RowLists& r) {
    FlatRows f;
    f.ptr.assign(r.size() + 1, 0);
    for (size_t i = 0; i < r.size(); ++i) f.ptr[i + 1] = f.ptr[i] + (int64_t)r[i].size();
    f.col.resize((size_t)f.ptr.back()); f.val.resize((size_t)f.ptr.back());
    for (size_t i = 0; i < r.size(); ++i)
        for (size_t u = 0; u < r[i].size(); ++u) { f.col[(size_t)f.ptr[i] + u] = r[i][u].first; f.val[(size_t)f.ptr[i] + u] = r[i][u].second; }
    return f;
}

// shared tail of every ILU-family setup: level-order both factors and hand out the preconditioner object
static int32_t finish_ilu_pc(kryst_csr_t a, int mode, bool divide, const FlatRows& le, const FlatRows& ue,
                             const std::vector<double>& dg, kryst_pc_t* out) {
    kryst_ctx_t ctx = a->ctx;
    const int64_t n = a->nrows;
    std::vector<double> ones((size_t)n, 1.0);
    kryst_pc_t pc = new kryst_pc_s();
    pc->ctx = ctx; pc->kind = KR_PC_ILU; pc->a = a; pc->n = n; pc->ilu_mode = mode; pc->divide_diag = divide;
    IluData* D = new IluData();
    D->n = n;
    pc->d_work = reinterpret_cast<double*>(D);
    std::vector<int32_t> posL, posU;
    int32_t rc = build_factor(n, le, ones, true, &D->L, &posL);
    if (rc == KRYST_OK) rc = build_factor(n, ue, dg, false, &D->U, &posU);
    if (rc == KRYST_OK) {
        std::vector<int32_t> mapLU((size_t)n);
        for (int64_t i = 0; i < n; ++i) mapLU[posU[i]] = posL[i];
        rc = up(&D->d_mapLU, mapLU);
    }
    if (rc == KRYST_OK && hipMalloc(&D->d_args, sizeof(TriArgs)) != hipSuccess) { set_error("hipMalloc failed"); rc = KRYST_ERR_HIP; }
    if (rc == KRYST_OK) {
        const size_t bytes = sizeof(double) * (size_t)((n + KR_TILE - 1) / KR_TILE * KR_TILE + KR_TILE);
        for (double** pp : {&D->d_y, &D->d_rL, &D->d_yU, &D->d_zU}) {
            if (rc != KRYST_OK) break;
            if (hipMalloc(pp, bytes) != hipSuccess) { set_error("hipMalloc failed"); rc = KRYST_ERR_HIP; }
            else if (hipMemsetAsync(*pp, 0, bytes, ctx->s_main) != hipSuccess) rc = KRYST_ERR_HIP;
        }
        if (rc == KRYST_OK && hipStreamSynchronize(ctx->s_main) != hipSuccess) rc = KRYST_ERR_HIP;
    }
    if (getenv("KRYST_ILU_VERBOSE"))
        fprintf(stderr, "[kryst ilu] n=%lld levels L/U=%zu/%zu\n", (long long)n, D->L.lvl_off.size() - 1, D->U.lvl_off.size() - 1);
    if (rc != KRYST_OK) { kryst_pc_destroy(pc); return rc; }
    *out = pc;
    return KRYST_OK;
}

static int32_t download_rows(kryst_csr_t a, std::vector<int64_t>& rp, std::vector<int32_t>& col, std::vector<double>& val) {
    rp.resize((size_t)a->nrows + 1); col.resize((size_t)a->nnz); val.resize((size_t)a->nnz);
    return kryst_csr_download(a, rp.data(), col.data(), val.data());
}


extern "C" int32_t kryst_pc_ilu0(kryst_csr_t a, int32_t mode, kryst_pc_t* out) {
    KR_ARG(a && out && mode >= 0 && mode <= 2, "pc_ilu0");
    KR_ARG(a->nrows == a->xlen, "pc_ilu0: square operator required");
    kryst_ctx_t ctx = a->ctx;
    KR_HIP(hipSetDevice(ctx->device));
    const int64_t n = a->nrows, nnz = a->nnz;
    std::vector<int64_t> rp((size_t)n + 1); std::vector<int32_t> col((size_t)nnz); std::vector<double> val((size_t)nnz);
    KR_TRY(kryst_csr_download(a, rp.data(), col.data(), val.data()));
    // diagonal position of every row (halo columns, col >= n, are outside the local block)
    std::vector<int64_t> dpos((size_t)n, -1);
    for (int64_t i = 0; i < n; ++i)
        for (int64_t k = rp[i]; k < rp[i + 1]; ++k) if (col[k] == i) dpos[i] = k;
    std::vector<double> w(val);                       // factor values on A's pattern
    if (mode == KRYST_ILU_TRUE_ILU0) {                // IKJ restricted to the pattern
        std::vector<int64_t> pos((size_t)n, -1);
        for (int64_t i = 0; i < n; ++i) {
            for (int64_t k = rp[i]; k < rp[i + 1]; ++k) if (col[k] < n) pos[col[k]] = k;
            for (int64_t k = rp[i]; k < rp[i + 1]; ++k) {
                const int64_t c = col[k];
                if (c >= i) break;
                const int64_t kd = dpos[c];
                if (kd < 0 || w[kd] == 0.0) { set_error("ILU(0): zero pivot at row %lld", (long long)c); return KRYST_ZERO_PIVOT; }
                w[k] = w[k] / w[kd];
                for (int64_t kk = rp[c]; kk < rp[c + 1]; ++kk) {
                    const int64_t j = col[kk];
                    if (j > c && j < n && pos[j] >= 0) w[pos[j]] = w[pos[j]] - w[k] * w[kk];
                }
            }
            for (int64_t k = rp[i]; k < rp[i + 1]; ++k) if (col[k] < n) pos[col[k]] = -1;
        }
    } else {
        // ilu.rs:76-80 / ilup.rs:104-111: l_ij = a_ij / a_jj for stored nonzeros below the diagonal; U = triu(A)
        for (int64_t i = 0; i < n; ++i)
            for (int64_t k = rp[i]; k < rp[i + 1]; ++k) {
                const int64_t j = col[k];
                if (j < i && val[k] != 0.0) {
                    const double ujj = dpos[j] >= 0 ? val[dpos[j]] : 0.0;
                    if (mode == KRYST_ILU_ILUP0 && ujj == 0.0) {           // ilup.rs:106-108
                        set_error("ILUP: zero diagonal in U at row %lld", (long long)j);
                        return KRYST_SOLVE_ERROR;
                    }
                    w[k] = val[k] / ujj;
                }
            }
    }
    const bool divide = mode != KRYST_ILU_KRYST_COMPAT;                    // ilu.rs:115-119 never divides
    FlatRows le, ue;
    le.ptr.assign((size_t)n + 1, 0); ue.ptr.assign((size_t)n + 1, 0);
    std::vector<double> dg((size_t)n, 1.0);
    for (int64_t i = 0; i < n; ++i) {                                      // count, then fill (stored order = ascending column)
        int64_t nl = 0, nu = 0;
        for (int64_t k = rp[i]; k < rp[i + 1]; ++k) {
            const int64_t j = col[k];
            if (j >= n || w[k] == 0.0) continue;                           // halo column / `!= T::zero()` filters
            if (j < i) ++nl; else if (j > i) ++nu;
        }
        le.ptr[i + 1] = le.ptr[i] + nl; ue.ptr[i + 1] = ue.ptr[i] + nu;
    }
    le.col.resize((size_t)le.ptr[n]); le.val.resize((size_t)le.ptr[n]); ue.col.resize((size_t)ue.ptr[n]); ue.val.resize((size_t)ue.ptr[n]);
    for (int64_t i = 0; i < n; ++i) {
        int64_t wl = le.ptr[i], wu = ue.ptr[i];
        for (int64_t k = rp[i]; k < rp[i + 1]; ++k) {
            const int64_t j = col[k];
            if (j >= n || w[k] == 0.0) continue;
            if (j < i) { le.col[wl] = (int32_t)j; le.val[wl] = w[k]; ++wl; }
            else if (j > i) { ue.col[wu] = (int32_t)j; ue.val[wu] = w[k]; ++wu; }
            else if (divide) dg[i] = w[k];                                 // ilup.rs:160-164 (missing diagonal: no divide)
        }
    }
    return finish_ilu_pc(a, mode, divide, le, ue, dg, out);
}

// Ilup::new(fill).setup(a) exactly as written (src/preconditioner/ilup.rs:77-134), on sparse rows instead of the reference's
// dense n x n `level` / `a_work` arrays: an entry that the dense code never touches is (0.0, usize::MAX) here too.
extern "C" int32_t kryst_pc_ilup(kryst_csr_t a, int32_t fill, kryst_pc_t* out) {
    KR_ARG(a && out && fill >= 0, "pc_ilup");
    KR_ARG(a->nrows == a->xlen, "pc_ilup: square operator required");
    if (fill == 0) return kryst_pc_ilu0(a, KRYST_ILU_ILUP0, out);
    KR_HIP(hipSetDevice(a->ctx->device));
    const int64_t n = a->nrows;
    std::vector<int64_t> rp; std::vector<int32_t> col; std::vector<double> val;
    KR_TRY(download_rows(a, rp, col, val));
    struct Ent { double v; uint64_t lev; };
    const uint64_t UMAX = ~0ull;
    struct UEnt { int32_t k; double v; uint64_t lev; };
    std::vector<std::vector<UEnt>> urows((size_t)n);       // every nonzero a_work[j][k], k > j, of a finished row j
    std::vector<double> udiag((size_t)n, 0.0);             // a_work[j][j] of a finished row
    RowLists le((size_t)n), ue((size_t)n);
    std::vector<double> dg((size_t)n, 1.0);
    std::map<int32_t, Ent> W;
    for (int64_t i = 0; i < n; ++i) {
        W.clear();
        for (int64_t k = rp[i]; k < rp[i + 1]; ++k)
            if (col[k] < n) W[col[k]] = Ent{val[k], val[k] != 0.0 ? 0ull : UMAX};            // ilup.rs:88-101 (halo columns dropped)
        for (auto it = W.begin(); it != W.end() && it->first < i; ++it) {                  // :104 `for j in 0..i`
            const int32_t j = it->first;
            const Ent ej = it->second;
            if (!(ej.v != 0.0 && ej.lev <= (uint64_t)fill)) continue;                      // :106
            const double u_jj = udiag[j];
            if (u_jj == 0.0) { set_error("ILUP: zero diagonal in U at row %d", j); return KRYST_SOLVE_ERROR; }   // :108-110
            const double lij = ej.v / u_jj;                                                // :112
            le[i].push_back({j, lij});
            for (const UEnt& u : urows[j]) {                                               // :116 `for k in (j+1)..n`
                uint64_t nl = ej.lev;                                                      // saturating adds (:118)
                nl = (nl > UMAX - u.lev) ? UMAX : nl + u.lev;
                nl = (nl == UMAX) ? UMAX : nl + 1;
                if (nl <= (uint64_t)fill) {
                    auto w = W.find(u.k);
                    if (w == W.end()) w = W.emplace(u.k, Ent{0.0, UMAX}).first;
                    const double update = lij * u.v;
                    w->second.v = w->second.v - update;                                    // :121
                    if (nl < w->second.lev) w->second.lev = nl;                            // :122
                }
            }
        }
        for (auto& kv : W) {
            if (kv.first < i) continue;
            if (kv.first == i) udiag[i] = kv.second.v;
            if (kv.second.v != 0.0 && kv.second.lev <= (uint64_t)fill) {                   // :129-134
                if (kv.first == i) dg[i] = kv.second.v;
                else ue[i].push_back({kv.first, kv.second.v});
            }
            if (kv.first > i && kv.second.v != 0.0) urows[i].push_back(UEnt{kv.first, kv.second.v, kv.second.lev});
        }
    }
    return finish_ilu_pc(a, 10 + fill, true, flatten(le), flatten(ue), dg, out);
}

// Ilut::new(fill, droptol).setup(a) exactly as written (src/preconditioner/ilut.rs:80-117): no elimination; drop by
// magnitude, keep the `fill` largest (stable descending sort), split at the diagonal.  Entries keep their STORED order
// (descending magnitude after a truncation), which is the order the apply subtracts them in (ilut.rs:129-141).
extern "C" int32_t kryst_pc_ilut(kryst_csr_t a, int32_t fill, double droptol, kryst_pc_t* out) {
    KR_ARG(a && out && fill >= 0, "pc_ilut");
    KR_ARG(a->nrows == a->xlen, "pc_ilut: square operator required");
    KR_HIP(hipSetDevice(a->ctx->device));
    const int64_t n = a->nrows;
    std::vector<int64_t> rp; std::vector<int32_t> col; std::vector<double> val;
    KR_TRY(download_rows(a, rp, col, val));
    RowLists le((size_t)n), ue((size_t)n);
    std::vector<double> dg((size_t)n, 1.0);
    std::vector<std::pair<int32_t, double>> row;
    for (int64_t i = 0; i < n; ++i) {
        row.clear();
        for (int64_t k = rp[i]; k < rp[i + 1]; ++k)
            if (col[k] < n && val[k] != 0.0 && std::fabs(val[k]) >= droptol) row.push_back({col[k], val[k]});     // :88-95
        if ((int64_t)row.size() > fill) {                                                                        // :97-100
            std::stable_sort(row.begin(), row.end(), [](const std::pair<int32_t, double>& x, const std::pair<int32_t, double>& y) {
                return std::fabs(x.second) > std::fabs(y.second); });
            row.resize((size_t)fill);
        }
        bool have_d = false;
        for (auto& e : row) {                                                                                    // :104-112
            if (e.first < i) le[i].push_back(e);
            else if (e.first > i) ue[i].push_back(e);
            else if (!have_d) { dg[i] = e.second; have_d = true; }                                               // :143-144
        }
    }
    return finish_ilu_pc(a, 100, true, flatten(le), flatten(ue), dg, out);
}
