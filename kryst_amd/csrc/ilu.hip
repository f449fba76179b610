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
#include "host_factor.h"
#include <chrono>
#include <thread>
#include <sched.h>
#include <sys/mman.h>
#include <cstdlib>
#include <new>
#include <pthread.h>
#include <memory>
#include <string>
#include <mutex>
#include <atomic>
#include "ew.h"
#include <algorithm>
#include <cmath>
#include <map>

namespace kr {

// (par_rows, the host block pool / hvec, the janitor thread, FlatRows and the host-side factorisations themselves: host_factor.h -- no device call
// in there, so that the same code builds for the CPU alone under the sanitizers)

// wavefront kernel by number of 8 x 8 line blocks in the (j, k) plane: the 16 x 16 kernel from 32^3 up (table at its use)
static int default_wave_form(unsigned nb8) { return nb8 >= 16 ? 2 : 1; }

#define KR_ILU_BOX_DEFAULT 2         // KRYST_ILU_BOX: 0 box-stencil factors take the level-ordered forms, 1 hyperplane launches, 2 pipelined wavefront (tri_box.h)

struct TriArgs {                    // device-resident argument block, rewritten before every apply (graph-friendly):
    const double* r; double* z; long long skip;   // one scalar load gives a level kernel everything it needs
    long long epoch;                              // number of this apply (tri_quad.h: the value of its "under way" flags)
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
    // operand streams of tri_run_free_kernel, built at set-up: per chunk of 64 level-order positions of a narrow run, [operand 0..7][lane] -- a
    // descriptor (> 0: the ring tag of an operand of this run that lies within reach, < 0: 0x80000000 | position of an operand to be gathered
    // from the vector, 0: no such entry) and the coefficient; what a lane needs lies where a coalesced load puts it, and nothing is classified per apply
    int32_t* d_fdesc = nullptr; double* d_fval = nullptr; int32_t* d_fpos = nullptr; int32_t* d_vreal = nullptr;   // (+ the operand's position in the vector; per virtual row its row and first / last flags)
    std::vector<int32_t> run_cbase; // first chunk of every narrow run in the streams, in the order enqueue_factor meets them
    std::vector<int32_t> run_nvirt; // virtual rows of every narrow run (a row of more than eight entries is a chain of them)
    int32_t last_entry = -1;        // index of the factor's last stored entry (tri_run_free_kernel clamps its look-ahead to it)
    bool free_runs = false;         // runs of narrow levels take tri_run_free_kernel (its 128 KiB of LDS were granted at set-up): the vector starts as sentinels
    int held = 16;                  // entries of a row the CSR sync-free kernel holds in registers (8: no row is longer than that)
    std::vector<int32_t> lvl_off;   // host: position offsets per level
    int32_t* d_lvl_off = nullptr;
    void free_all() { (void)pool_free(d_ptr); (void)pool_free(d_col); (void)pool_free(d_val); (void)pool_free(d_row); (void)pool_free(d_diag); (void)pool_free(d_lvl_off);
                      (void)pool_free(d_fdesc); (void)pool_free(d_fval); (void)pool_free(d_fpos); (void)pool_free(d_vreal);
                      (void)pool_free(d_ecol); (void)pool_free(d_eval); (void)pool_free(d_elen); }
    EllView view() const { return EllView{d_ecol, d_eval, d_elen, npos}; }
};

// Structured-grid form of one factor (natural row order): every kept entry of row (i,j,k) of an Ni x Nj x Nk box couples it to
// its -1 / -Ni / -Ni*Nj neighbour (L) or +1 / +Ni / +Ni*Nj neighbour (U) -- any ILU-family factor of a 7-point (or 5-point)
// operator in natural ordering.  Such a factor is solved by the pipelined wavefront kernel tri_grid_kernel instead of the
// level machinery.
struct GridFactor {
    bool ok = false;
    int32_t Ni = 0, Nj = 0, Nk = 0;
    double* d_c1 = nullptr; double* d_c2 = nullptr; double* d_c3 = nullptr;   // coefficient of the i / j / k neighbour
                                                                            // (0.0 = no such entry: zero entries are never stored)
    double* d_diag = nullptr;                                               // divisor (backward factor only)
    // blocked copy for tri_quad_kernel (16 x 16 lines per workgroup): [block][chunk][array][step pair][line], built on the device
    void* d_blocked = nullptr; int32_t nbj = 0, nbk = 0, nch = 0;
    double* d_edge_e = nullptr; double* d_edge_n = nullptr;                // edge rows handed to the next workgroup: [block][steps + 8][16]
    uint8_t* d_skip = nullptr;                                              // [block][quadrant][chunk]: coefficients repeat chunk - 3's (tri_quad_dedup_kernel)
    int64_t nskip = 0;                                                      // how many of them do (KRYST_ILU_VERBOSE)
    void free_all() { (void)pool_free(d_c1); (void)pool_free(d_c2); (void)pool_free(d_c3); (void)pool_free(d_diag);
                      (void)pool_free(d_blocked); (void)pool_free(d_edge_e); (void)pool_free(d_edge_n); (void)pool_free(d_skip); }
};
struct GridView { int32_t Ni, Nj, Nk; const double* c1; const double* c2; const double* c3; const double* diag; };

// Box-stencil form of one factor (round 4): every kept entry of row (i, j, k) of an Ni x Nj x Nk box couples it to (i + di, j + dj, k + dk)
// with |di|, |dj|, |dk| <= 1 -- strictly lower (L) or strictly upper (U) in natural ordering: any ILU-family factor of a stencil inside the
// 3 x 3 x 3 cube (27-point, 19-point, 9-point 2-D, ...) that is not a 7-point factor (those take GridFactor).  Thirteen coefficient streams
// in natural row order, stream a = the a-th offset in ASCENDING COLUMN order (the order the reference subtracts in): L: (dk, dj, di) =
// (-1,-1,-1), (-1,-1,0), ... (0,0,-1), i.e. a = 9 (dk + 1) + 3 (dj + 1) + (di + 1); U: (0,0,1), (0,1,-1), ... (1,1,1), a = that code - 14.
// 0.0 = no such entry (zero entries are never stored).
struct BoxFactor {
    bool ok = false;
    int32_t Ni = 0, Nj = 0, Nk = 0;
    double* d_c = nullptr;          // 13 streams, box_stream_stride(n) doubles apart: d_c[a stride + row]
    double* d_diag = nullptr;       // divisor (backward factor only)
    void* d_cb = nullptr;           // blocked copy for tri_box_kernel: [block][chunk][stream (, divisor)][step pair][lane] (tri_box_layout_kernel)
    uint32_t present = 0x1fff;      // streams with at least one entry
    bool regular = false;           // every present stream has an entry wherever the neighbour row exists in the box (tri_box.h: REGULAR)
    void free_all() { (void)pool_free(d_c); (void)pool_free(d_diag); (void)pool_free(d_cb); d_c = nullptr; d_diag = nullptr; d_cb = nullptr; }
};
struct BoxView { int32_t Ni, Nj, Nk; int64_t n; const double* c; const double* diag; int64_t cs; const void* cb; };   // cs: doubles from one stream to the next; cb: blocked copy (tri_box.h)
// Streams are NOT n doubles apart: with n = 128^3 that is 16 MiB, and the 13 coefficients of a row would sit in the same HBM channel and bank
// (measured: 27-point 128^3 apply 1.75 ms where the hop / step model says 1.0).  n rounded up to 64 rows plus 72 rows: consecutive streams are
// 576 bytes apart modulo any power of two from 1 KiB up.
static inline int64_t box_stream_stride(int64_t n) { return (n + 63) / 64 * 64 + 72; }

struct IluData {
    TriFactor L, U;
    GridFactor GL, GU;
    BoxFactor BL, BU;
    TriArgs* d_args = nullptr;
    double* d_rL = nullptr; double* d_y = nullptr; double* d_yU = nullptr; double* d_zU = nullptr;   // level-permuted work vectors
    int32_t* d_mapLU = nullptr;     // L-position of the row at U-position q
    int32_t* d_flags = nullptr;     // wavefront solve: "this block is under way", one per block and direction
    int32_t direct_epoch = 0;       // number of the last directly launched apply (1 .. 2^30 - 1; its flag value has bit 30 set)
    int32_t* h_gave_up = nullptr;   // mapped host word the wavefront kernel raises when a poller runs out of patience (never cleared on the device)
    int32_t* d_gave_up = nullptr;   // its device address
    bool box_wave_ready = false;    // box factor: flags, give-up word and the kernels' LDS size are set up for tri_box_kernel
    bool safe = false;              // the wavefront kernel gave up once: this preconditioner now uses the plane kernels (no inter-workgroup waits)
    bool fell_back = false;         // ... and the switch happened since the last pc_fell_back() query
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
    // (all (<= 4) dependencies are polled every round: gating them on the deepest one, which pays for rows of 13 entries in
    // tri_syncfree_csr_kernel, adds a dependent round trip here -- 256^3: 3.1 -> 4.2 ms)
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

// the same for factors with longer rows (CSR in level order): a lane advances through its entries as far as they are ready.
// The row's first P entries (columns AND values) are in registers before the lane starts waiting: once its gate opens, what stands
// between the row and its result is ONE uncached round trip for all of them -- with the entries fetched after the gate (round 2:
// batches of eight, columns from L2, values a cold miss each) a 27-point factor's 13 entries per row cost 7.4 us per dependency level.
template <bool FORWARD, int P>
__global__ __launch_bounds__(256) void tri_syncfree_csr_kernel(const TriArgs* args, const double* __restrict__ in, double* out,
                                                               const int32_t* __restrict__ ptr, const int32_t* __restrict__ col,
                                                               const double* __restrict__ val, const double* __restrict__ diag, int32_t npos) {
    if (args->skip) return;
    const int32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    const bool active = p < npos;
    int32_t k = 0, kend = 0; double s = 0.0, dg = 1.0;
    if (active) { k = ptr[p]; kend = ptr[p + 1]; s = in[p]; if (!FORWARD) dg = diag[p]; }
    bool done = !active;
    int32_t cc[P]; double vv[P];
#pragma unroll
    for (int u = 0; u < P; ++u) {
        const int32_t kk = min(k + u, kend - 1);
        cc[u] = k < kend ? col[kk] : 0;
        vv[u] = k < kend ? val[kk] : 0.0;
    }
    // While a row waits it polls ONE entry: the dependency at the largest level-order position, i.e. of the deepest level -- the
    // last one to be solved in all but rare cases (eight uncached loads per waiting lane and round slowed everybody down: 20 -> 40 ms
    // on a 27-point factor).  Once that one is there, the batch below usually finds everything ready.
    int32_t gate_col = -1;
#pragma unroll
    for (int u = 0; u < P; ++u) if (k + u < kend) gate_col = max(gate_col, cc[u]);
    for (int32_t kk = k + P; kk < kend; ++kk) gate_col = max(gate_col, col[kk]);
    bool open = gate_col < 0;
    bool first = true;
    for (int budget = 1 << 22; budget > 0; --budget) {
        if (!done && !open) {
            const double g = __hip_atomic_load(&out[gate_col], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            open = (unsigned long long)__double_as_longlong(g) != KR_TRI_SENTINEL;
        }
        if (!done && open) {
            if (first) {
                // the register-held entries: asked for TOGETHER, the ready prefix consumed in stored order
                first = false;
                double xv[P];
#pragma unroll
                for (int u = 0; u < P; ++u) xv[u] = k + u < kend ? __hip_atomic_load(&out[cc[u]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
                bool prefix = true;
                int adv = 0;
#pragma unroll
                for (int u = 0; u < P; ++u) {
                    const bool ready = (unsigned long long)__double_as_longlong(xv[u]) != KR_TRI_SENTINEL;
                    prefix = prefix && k + u < kend && ready;
                    if (prefix) { s = s - vv[u] * xv[u]; ++adv; }                          // stored order
                }
                k += adv;
            } else {
                // what is left (a gate that opened before an earlier dependency, rows longer than P): batches of eight from memory
                constexpr int B = 8;
                double xv[B], wv[B];
#pragma unroll
                for (int u = 0; u < B; ++u) {
                    const int32_t kk = min(k + u, kend - 1);
                    wv[u] = k < kend ? val[kk] : 0.0;
                    xv[u] = k < kend ? __hip_atomic_load(&out[col[kk]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
                }
                bool prefix = true;
                int adv = 0;
#pragma unroll
                for (int u = 0; u < B; ++u) {
                    const bool ready = (unsigned long long)__double_as_longlong(xv[u]) != KR_TRI_SENTINEL;
                    prefix = prefix && k + u < kend && ready;
                    if (prefix) { s = s - wv[u] * xv[u]; ++adv; }                          // stored order
                }
                k += adv;
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

// PIPELINED WAVEFRONT solve of a structured-grid factor (GridFactor).  One wave per workgroup owns an 8 x 8 block of grid
// lines (j,k); lane (jl,kl) walks its line i = 0..Ni-1 (backward: everything mirrored), one row per step, skewed by jl + kl
// steps, so that when it reaches row i its j- and k-neighbours' row i was finished one step earlier by the lanes next to it:
// those two values travel by wave shuffles, the i-neighbour stays in a register.  Only block-boundary lanes read neighbours'
// results from memory (another workgroup's output, agent-scope loads, NaN sentinel = not yet written) and they request them
// P steps ahead, so in steady state nobody waits: a dependency hop costs a step of register/shuffle work (~0.1 us) instead of
// a memory round trip (~2 us in the level-order sync-free kernel).  Blocks are dispatched in (K, J) order, so a block only
// waits for lower-numbered ones.  The subtraction order is the stored (ascending column) order of the row: bit-identical.
// Rows are fetched in CHUNKS of 8 steps with 16-byte loads, one chunk ahead (a lane's rows are consecutive in memory), so a
// step issues no load of its own; an entry is present iff its coefficient is nonzero (zero entries are never stored).
template <bool FORWARD>
__global__ __launch_bounds__(64) void tri_grid_kernel(const TriArgs* args, const double* in_ptr, double* out_ptr, GridView G, int64_t n) {
    if (args->skip) return;
    constexpr int C = 8;
    const double* __restrict__ in = in_ptr ? in_ptr : args->r;
    double* out = out_ptr ? out_ptr : args->z;
    const int l = threadIdx.x, jl = l & 7, kl = l >> 3, skew = jl + kl;
    const int nbj = (G.Nj + 7) >> 3;
    const int J = blockIdx.x % nbj, K = blockIdx.x / nbj;
    const int jj = J * 8 + jl, kk = K * 8 + kl;                           // schedule coordinates (mirrored for the backward solve)
    const bool line_ok = jj < G.Nj && kk < G.Nk;
    const int j = FORWARD ? jj : G.Nj - 1 - jj, k = FORWARD ? kk : G.Nk - 1 - kk;
    const int32_t s1 = G.Ni, s2 = G.Ni * G.Nj;
    const int64_t line0 = line_ok ? (int64_t)(k * G.Nj + j) * G.Ni : 0;   // row of i = 0 on this line
    const int32_t dj = FORWARD ? -s1 : s1, dk = FORWARD ? -s2 : s2;       // where the j / k neighbour's row lives
    const bool west_glob = line_ok && jl == 0 && J > 0, south_glob = line_ok && kl == 0 && K > 0;
    auto is_sentinel = [](double x) { return (unsigned long long)__double_as_longlong(x) == KR_TRI_SENTINEL; };
    auto row_of = [&](int ii) -> int64_t { return line0 + (FORWARD ? ii : G.Ni - 1 - ii); };   // may lie outside the line (unused then)
    struct Chunk { double rv[C], a1[C], a2[C], a3[C], dg[C], wv[C], sv[C]; };
    // the C rows of steps t .. t+C-1 are contiguous: ascending for the forward solve, descending for the backward one
    auto fetch = [&](Chunk& q, int t) {
        const int ii0 = t - skew;
        const int64_t lo = FORWARD ? row_of(ii0) : row_of(ii0 + C - 1);  // lowest row of the chunk
        if (lo >= 0 && lo + C <= n) {
            typedef double v2 __attribute__((ext_vector_type(2)));
#pragma unroll
            for (int h = 0; h < C / 2; ++h) {
                const int e0 = FORWARD ? 2 * h : C - 1 - 2 * h, e1 = FORWARD ? 2 * h + 1 : C - 2 - 2 * h;   // step index of the pair's elements
                const v2 r2 = *reinterpret_cast<const v2*>(in + lo + 2 * h);
                const v2 x1 = *reinterpret_cast<const v2*>(G.c1 + lo + 2 * h), x2 = *reinterpret_cast<const v2*>(G.c2 + lo + 2 * h);
                const v2 x3 = *reinterpret_cast<const v2*>(G.c3 + lo + 2 * h);
                q.rv[e0] = r2.x; q.rv[e1] = r2.y; q.a1[e0] = x1.x; q.a1[e1] = x1.y; q.a2[e0] = x2.x; q.a2[e1] = x2.y; q.a3[e0] = x3.x; q.a3[e1] = x3.y;
                if (!FORWARD) { const v2 d2v = *reinterpret_cast<const v2*>(G.diag + lo + 2 * h); q.dg[e0] = d2v.x; q.dg[e1] = d2v.y; }
            }
        } else {                                                           // first / last rows of the whole vector: element-wise, clamped
#pragma unroll
            for (int u = 0; u < C; ++u) {
                const int64_t row = min(max(row_of(ii0 + u), (int64_t)0), n - 1);
                q.rv[u] = in[row]; q.a1[u] = G.c1[row]; q.a2[u] = G.c2[row]; q.a3[u] = G.c3[row];
                if (!FORWARD) q.dg[u] = G.diag[row];
            }
        }
#pragma unroll
        for (int u = 0; u < C; ++u) {                                       // neighbour blocks' results (boundary lanes only)
            const int ii = ii0 + u;
            const bool in_line = ii >= 0 && ii < G.Ni;
            q.wv[u] = (west_glob && in_line) ? __hip_atomic_load(&out[row_of(ii) + dj], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
            q.sv[u] = (south_glob && in_line) ? __hip_atomic_load(&out[row_of(ii) + dk], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
        }
    };
    double y_own = 0.0, y_last = 0.0;
    auto run = [&](Chunk& q, int t0) {
#pragma unroll
        for (int u = 0; u < C; ++u) {
            const int ii = t0 + u - skew;
            const bool act = line_ok && ii >= 0 && ii < G.Ni;
            const int64_t row = row_of(ii);
            double yj = __shfl_up(y_last, 1, 64), yk = __shfl_up(y_last, 8, 64);
            // block-boundary lanes: a value requested a chunk ago that had not been written yet is requested again, now
            bool need_w = act && west_glob && q.a2[u] != 0.0 && is_sentinel(q.wv[u]);
            bool need_s = act && south_glob && q.a3[u] != 0.0 && is_sentinel(q.sv[u]);
            for (int budget = 1 << 22; __any(need_w || need_s) && budget > 0; --budget) {
                if (need_w) { q.wv[u] = __hip_atomic_load(&out[row + dj], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); need_w = is_sentinel(q.wv[u]); }
                if (need_s) { q.sv[u] = __hip_atomic_load(&out[row + dk], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); need_s = is_sentinel(q.sv[u]); }
                __builtin_amdgcn_s_sleep(1);
            }
            if (west_glob) yj = q.wv[u];
            if (south_glob) yk = q.sv[u];
            double s = q.rv[u];
            if (FORWARD) {                                                // stored order: k-, j-, i-neighbour (ascending column)
                if (q.a3[u] != 0.0) s = s - q.a3[u] * yk;
                if (q.a2[u] != 0.0) s = s - q.a2[u] * yj;
                if (q.a1[u] != 0.0) s = s - q.a1[u] * y_own;
            } else {                                                      // i-, j-, k-neighbour, then the divisor
                if (q.a1[u] != 0.0) s = s - q.a1[u] * y_own;
                if (q.a2[u] != 0.0) s = s - q.a2[u] * yj;
                if (q.a3[u] != 0.0) s = s - q.a3[u] * yk;
                s = s / q.dg[u];
            }
            if (act) {
                y_own = s; y_last = s;
                __hip_atomic_store(&out[row], s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    };
    Chunk qa, qb;
    const int nsteps = G.Ni + 14;
    fetch(qa, 0);
    for (int t0 = 0; t0 < nsteps; t0 += 2 * C) {                          // two chunks per trip: the buffers swap roles without copies
        fetch(qb, t0 + C);
        run(qa, t0);
        fetch(qa, t0 + 2 * C);
        run(qb, t0 + C);
    }
}

}  // namespace kr
#include "tri_wave.h"
#include "tri_quad.h"
#include "tri_box.h"
namespace kr {

__global__ __launch_bounds__(256) void tri_fill_kernel(const TriArgs* args, double* dst_ptr, int64_t n, int32_t* flags = nullptr, int32_t nflags = 0) {
    if (args->skip) return;
    if (flags && (int64_t)blockIdx.x * blockDim.x + threadIdx.x < nflags) flags[(int64_t)blockIdx.x * blockDim.x + threadIdx.x] = 0;
    double* dst = dst_ptr ? dst_ptr : args->z;
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p < n) dst[p] = __longlong_as_double((long long)KR_TRI_SENTINEL);
}

// Structured-grid factor, one launch per hyperplane i + j + k = level (schedule coordinates; mirrored for the backward solve):
// no workgroup waits for another, so it needs no assumption about dispatch order.  ~4 us per level: the fallback of the
// wavefront kernel (tri_wave.h), never the default.  Same subtraction order, same bits.
template <bool FORWARD>
__global__ __launch_bounds__(256) void tri_plane_kernel(const TriArgs* args, const double* in_ptr, double* out_ptr, GridView G, int level) {
    if (args->skip) return;
    const double* in = in_ptr ? in_ptr : args->r;
    double* out = out_ptr ? out_ptr : args->z;
    const int jk = blockIdx.x * blockDim.x + threadIdx.x;
    if (jk >= G.Nj * G.Nk) return;
    const int jj = jk % G.Nj, kk = jk / G.Nj, ii = level - jj - kk;
    if (ii < 0 || ii >= G.Ni) return;
    const int i = FORWARD ? ii : G.Ni - 1 - ii, j = FORWARD ? jj : G.Nj - 1 - jj, k = FORWARD ? kk : G.Nk - 1 - kk;
    const int64_t s1 = G.Ni, s2 = (int64_t)G.Ni * G.Nj;
    const int64_t row = i + s1 * j + s2 * k;
    const double a1 = G.c1[row], a2 = G.c2[row], a3 = G.c3[row];      // 0.0 = no such entry
    double s = in[row];
    if (FORWARD) {                                                     // stored order: k-, j-, i-neighbour (ascending column)
        if (a3 != 0.0) s = s - a3 * out[row - s2];
        if (a2 != 0.0) s = s - a2 * out[row - s1];
        if (a1 != 0.0) s = s - a1 * out[row - 1];
    } else {
        if (a1 != 0.0) s = s - a1 * out[row + 1];
        if (a2 != 0.0) s = s - a2 * out[row + s1];
        if (a3 != 0.0) s = s - a3 * out[row + s2];
        s = s / G.diag[row];
    }
    out[row] = s;
}

// Box-stencil factor, one launch per hyperplane i + 2 j + 4 k = h (schedule coordinates; mirrored for the backward solve): every lower
// offset (dk, dj, di) of the 3 x 3 x 3 cube has di + 2 dj + 4 dk <= -1 -- the closest are (0, -1, +1) and (-1, +1, +1) -- so the rows of a
// plane only need rows of earlier planes.  Ni + 2 Nj + 4 Nk - 6 launches per factor (replayed as a hipGraph): the barrier-free form of the
// box solve, and the one a preconditioner falls back to.  Stored (ascending column) order, absent entries skipped: the reference's bits.
template <bool FORWARD>
__global__ __launch_bounds__(256) void tri_box_plane_kernel(const TriArgs* args, const double* in_ptr, double* out_ptr, BoxView B, int h) {
    if (args->skip) return;
    const double* in = in_ptr ? in_ptr : args->r;
    double* out = out_ptr ? out_ptr : args->z;
    const int jk = blockIdx.x * blockDim.x + threadIdx.x;
    if (jk >= B.Nj * B.Nk) return;
    const int jj = jk % B.Nj, kk = jk / B.Nj, ii = h - 2 * jj - 4 * kk;
    if (ii < 0 || ii >= B.Ni) return;
    const int i = FORWARD ? ii : B.Ni - 1 - ii, j = FORWARD ? jj : B.Nj - 1 - jj, k = FORWARD ? kk : B.Nk - 1 - kk;
    const int64_t s1 = B.Ni, s2 = (int64_t)B.Ni * B.Nj;
    const int64_t row = i + s1 * j + s2 * k;
    double s = in[row];
#pragma unroll
    for (int a = 0; a < 13; ++a) {
        const int code = FORWARD ? a : a + 14;                             // 9 (dk + 1) + 3 (dj + 1) + (di + 1)
        const int dk = code / 9 - 1, dj = (code / 3) % 3 - 1, di = code % 3 - 1;
        const double c = B.c[(int64_t)a * B.cs + row];
        if (c != 0.0) s = s - c * out[row + di + s1 * dj + s2 * dk];
    }
    out[row] = FORWARD ? s : s / B.diag[row];
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

// The same run, software-pipelined: a level of a narrow factor is three DEPENDENT round trips (row pointers -> entries -> solution
// values) and a workgroup barrier, 3.4 us; only the last of the three depends on the level before.  Here the row pointers and the
// right-hand side of level lv + 2 and the first H entries (columns and values) of the rows of level lv + 1 are requested BEFORE level lv
// is computed, so that between two barriers there is one gather of solution values (this workgroup wrote them: L2 at worst), the
// subtractions in stored order and the store.  Each thread owns up to two rows of a level (levels of at most 2 048 rows, 1 024
// threads: the pipelined row is the thread's first one, a second one is done as in tri_run_kernel); rows longer than H continue from
// memory.  The same operations in the same order as tri_run_kernel: same bits.
template <bool FORWARD, int H>
__global__ __launch_bounds__(1024) void tri_run_pipe_kernel(const TriArgs* args, const double* __restrict__ in, double* out,
                                                            const int32_t* __restrict__ ptr, const int32_t* __restrict__ col,
                                                            const double* __restrict__ val, const double* __restrict__ diag,
                                                            const int32_t* __restrict__ lvl_off, int32_t l0, int32_t l1) {
    if (args->skip) return;

    struct Head { int32_t p, k0, k1; double s, dg; };                      // a row's position (-1: none), entry range, right-hand side, divisor
    struct Body { int32_t c[H]; double v[H]; };
    auto head_of = [&](int32_t lv) -> Head {
        Head h{-1, 0, 0, 0.0, 1.0};
        if (lv < l1) {
            const int32_t p = lvl_off[lv] + (int32_t)threadIdx.x;
            if (p < lvl_off[lv + 1]) { h.p = p; h.k0 = ptr[p]; h.k1 = ptr[p + 1]; h.s = in[p]; if (!FORWARD) h.dg = diag[p]; }
        }
        return h;
    };
    auto body_of = [&](const Head& h) -> Body {
        Body b;
#pragma unroll
        for (int u = 0; u < H; ++u) {
            const bool on = h.p >= 0 && h.k0 + u < h.k1;
            b.c[u] = on ? col[h.k0 + u] : 0;
            b.v[u] = on ? val[h.k0 + u] : 0.0;
        }
        return b;
    };
    Head hc = head_of(l0), hn = head_of(l0 + 1);
    Body bc = body_of(hc);
    for (int32_t lv = l0; lv < l1; ++lv) {
        // vector loads return in order: the gather of THIS level is requested first, the look-ahead loads behind it, so that the
        // subtractions wait for the gather alone (look-ahead first would put its cold misses back on the level's critical path)
        double x[H];
#pragma unroll
        for (int u = 0; u < H; ++u) x[u] = out[bc.c[u]];                              // (slots past the row's end read position 0: valid, unused)
        asm volatile("" ::: "memory");
        const Body bn = body_of(hn);                                       // in flight while this level is computed
        const Head hnn = head_of(lv + 2);
        asm volatile("" ::: "memory");
        if (hc.p >= 0) {
            double s = hc.s;
#pragma unroll
            for (int u = 0; u < H; ++u) if (hc.k0 + u < hc.k1) s = s - bc.v[u] * x[u];    // stored order
            for (int32_t k = hc.k0 + H; k < hc.k1; ++k) s = s - val[k] * out[col[k]];
            out[hc.p] = FORWARD ? s : s / hc.dg;
        }
        // levels wider than the workgroup (up to 2 048 rows): the rest as in tri_run_kernel
        for (int32_t p = lvl_off[lv] + (int32_t)threadIdx.x + (int32_t)blockDim.x; p < lvl_off[lv + 1]; p += blockDim.x) {
            double s = in[p];
            for (int32_t k = ptr[p]; k < ptr[p + 1]; ++k) s = s - val[k] * out[col[k]];
            out[p] = FORWARD ? s : s / diag[p];
        }
        __syncthreads();
        hc = hn; bc = bn; hn = hnn;
    }
}

// The same run WITHOUT barriers (round 4, late): one workgroup, its waves take the run's positions in chunks of 64, round-robin, and a lane
// waits for the rows ITS row needs instead of for the whole level.  A solved value is published twice: to the solution vector (agent-scope
// store; the vector starts as sentinels, perm_kernel) and to an LDS ring of the last TRF_RING positions as two self-validating 8-byte words
// (32 bits of the value | the position's tag).  EVERY operand of a row is an LDS word pair with an expected tag: a ring slot for a row of this
// run that no wave can have overtaken yet, a constant pair for an absent entry (value +0.0, coefficient set to 0.0: s - 0.0 * 0.0 == s bit for
// bit), a pair private to the lane for an operand from before the run or further back -- gathered from the vector at the chunk's start, ahead
// of the look-ahead loads (vector loads return in order).  So a turn of the loop has no branch per operand: read the eight pairs, compare the
// sixteen tags, and if all agree subtract in stored order and publish.  Three forms of that loop, tried in this order per chunk: LEAN (every
// unsolved lane looks every turn; ~120 instructions, and a lone wave issues one per ~4.5 cycles, so that is the latency of a dependent hop),
// GATED (a lane first polls the pair of its operand at the largest position; passes batched over the wave's lanes) and GENERAL (vector loads
// allowed: rows longer than H, operands not yet in the vector, ring slots reused under a reader -- a LATER tag).  tri_run_kernel's operations
// in tri_run_kernel's order.  Progress: the lowest unsolved position's row needs only solved rows, and the wave that owns its chunk has finished
// all its earlier chunks (all positions below), so it is on that chunk now; every wave of the one workgroup is resident.  Positions that share
// a ring slot belong to one wave (TRF_RING / 64 is a multiple of the wave count) and are solved in order.  The look-ahead: a chunk's operand
// streams (descriptor and coefficient per operand and lane, laid out at set-up: trf_stream_kernel) are requested two chunks ahead with 16 coalesced
// loads, the gathers from the vector one chunk ahead.  Rows of more than eight entries are chains of virtual rows (below).  The waves are held
// within TRF_AHEAD chunks of each other, so that no ring slot is reused under a reader.  A poll budget turns a logic error into NaNs instead of
// a hung GPU.
#define TRF_RING 4096
#define TRF_THREADS 512
#define TRF_AHEAD 8
#define TRF_LDS_BYTES (TRF_RING * 16 + TRF_THREADS * 8 * 16 + 16 + 64)
template <bool FORWARD, int H>
__global__ __launch_bounds__(TRF_THREADS) void tri_run_free_kernel(const TriArgs* args, const double* __restrict__ in, double* out,
                                                                   const double* __restrict__ diag, const int32_t* __restrict__ vreal,
                                                                   const int32_t* __restrict__ fdesc, const double* __restrict__ fval,
                                                                   const int32_t* __restrict__ fpos, int32_t cbase, int32_t NV, int tune) {
    // Positions are VIRTUAL rows of this run, 0 .. NV: a row of at most eight entries is one virtual row; a longer one is a chain of them -- the
    // first with eight operands, each further one with the partial sum of the one before as operand 0 (taken over, not subtracted) and seven
    // more -- and only the last of a chain divides and stores to the vector.  Same subtractions in the same order (trf_stream_kernel).
    static_assert(H == 8, "eight private pairs per lane");
    if (args->skip) return;
    extern __shared__ unsigned long long trf_lds[];                      // [TRF_RING][2] ring, [TRF_THREADS][8][2] private pairs, [2] the constant pair
    constexpr uint32_t PRIV = TRF_RING * 2, CONSTP = PRIV + TRF_THREADS * 16;      // (in words)
    constexpr uint32_t TAG_PRIV = 0xFFFFFFFEu, TAG_CONST = 0xFFFFFFFFu;
    for (int x = threadIdx.x; x < (int)CONSTP; x += blockDim.x) trf_lds[x] = 0ull;         // tag 0: never written (LDS is not cleared between kernels)
    if (threadIdx.x == 0) { trf_lds[CONSTP] = (unsigned long long)TAG_CONST << 32; trf_lds[CONSTP + 1] = (unsigned long long)TAG_CONST << 32; }
    const int W = blockDim.x >> 6, w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int32_t nchunk = (NV + 63) >> 6;
    // Which chunk every wave is on (all its earlier ones are finished): a wave starts chunk c only when no wave is more than TRF_AHEAD chunks
    // behind it.  An operand is taken from the ring up to `reach` = 55 chunks back (trf_stream_kernel), a slot is reused 64 chunks on: with
    // 55 + TRF_AHEAD < 64 no pair is overwritten while a row that reads it is unsolved.  (Without it a wave whose rows need nothing of a slow
    // wave's chunk ran 64 chunks ahead on a 27-point box and overwrote the partial sum a chain was waiting for.)
    int* const wave_at = reinterpret_cast<int*>(&trf_lds[CONSTP + 2]);
    if (lane == 0) __hip_atomic_store(&wave_at[w], w < nchunk ? w : INT32_MAX, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __syncthreads();
    const int batch = tune & 255;

    struct Head { int32_t p, vp; bool first, last; double s, dg; };       // p: the row's position in the vector (-1: no row), vp: virtual position
    struct Stage { int32_t d[H]; double v[H]; };                           // a chunk's operand streams: descriptor and coefficient of operand u of this lane's row
    // every load of the look-ahead is issued by every lane (clamped indices): the waits can then be COUNTED (s_waitcnt vmcnt(n))
    auto head_of = [&](int32_t ch) -> Head {
        Head h;
        const int32_t cc = ch < nchunk ? ch : nchunk - 1;
        const int32_t vr = *reinterpret_cast<const int32_t*>(reinterpret_cast<const char*>(vreal) + ((uint32_t)(cbase + cc) * 64u + (uint32_t)lane) * 4u);
        const bool valid = ch < nchunk && vr != 0;
        const uint32_t pc = valid ? (uint32_t)(vr & 0x1FFFFFFF) - 1u : 0u;
        h.s = *reinterpret_cast<const double*>(reinterpret_cast<const char*>(in) + pc * 8u);
        h.dg = FORWARD ? 1.0 : *reinterpret_cast<const double*>(reinterpret_cast<const char*>(diag) + pc * 8u);
        h.p = valid ? (int32_t)pc : -1;
        h.vp = (ch << 6) + lane;
        h.first = (vr >> 30) & 1; h.last = (vr >> 29) & 1;
        return h;
    };
    // 16 fully coalesced loads (32-bit byte offsets from the uniform base: one address instruction per load).  (With each lane loading ITS row's
    // entries from the CSR arrays -- the pipe kernel's way -- every cache line is asked for by eight different instructions: 650 L1 accesses per
    // chunk, the L1 busy or stalled on pending lines for 60 % of the run; PMC, round 4.)
    auto stage_of = [&](int32_t ch, Stage& st) {
        const uint32_t base = (uint32_t)(cbase + (ch < nchunk ? ch : nchunk - 1)) * (uint32_t)(H * 64) + (uint32_t)lane;
#pragma unroll
        for (int u = 0; u < H; ++u) {
            st.d[u] = *reinterpret_cast<const int32_t*>(reinterpret_cast<const char*>(fdesc) + (base + 64u * u) * 4u);
            st.v[u] = *reinterpret_cast<const double*>(reinterpret_cast<const char*>(fval) + (base + 64u * u) * 8u);
        }
    };
    auto pair_at = [&](uint32_t word, unsigned long long& a, unsigned long long& b) {
        a = __hip_atomic_load(&trf_lds[word], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        b = __hip_atomic_load(&trf_lds[word + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    };
    auto put_pair = [&](uint32_t word, double x, uint32_t tag) {
        const unsigned long long bits = (unsigned long long)__double_as_longlong(x), t = (unsigned long long)tag << 32;
        __hip_atomic_store(&trf_lds[word], (bits & 0xffffffffull) | t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_store(&trf_lds[word + 1], (bits >> 32) | t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    };
    auto publish = [&](const Head& h, double res) {                        // (a partial sum only to the ring: its one reader is the next virtual row)
        if (h.last) __hip_atomic_store(&out[h.p], res, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        put_pair(((uint32_t)h.vp & (TRF_RING - 1)) * 2, res, (uint32_t)h.vp + 1u);
    };
    auto solved = [](double x) { return (unsigned long long)__double_as_longlong(x) != KR_TRI_SENTINEL; };
    // one chunk: its head hc and its operand streams bc are here; the next chunk's streams (into st) and the head after that are requested on the way
    auto chunk = [&](const Head& hc, const Stage& bc, const Stage& nx, Stage& st, Head& hnn, int32_t ch, double (&xg)[H]) {
        const int len = H, nh = H;                                         // (rows longer than H are chains of virtual rows: nothing is left over)
        if (lane == 0) __hip_atomic_store(&wave_at[w], ch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        for (int spins = 0; spins < (1 << 24); ++spins) {
            const int at = __hip_atomic_load(&wave_at[lane < W ? lane : 0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (__all(ch - at <= TRF_AHEAD)) break;
            __builtin_amdgcn_s_sleep(2);
        }
        // every operand as (word offset of its pair, expected tag); bit u of `vecm`: operand u comes from the vector, of `vpend`: ... and is not there yet
        uint32_t word[H], etag[H];
        unsigned vecm = 0, vpend = 0;
        int32_t gpos = -1;                                                 // (the largest ring tag among the operands)
        // the operand's position in the vector: in its descriptor, or -- for a ring operand, asked for only when its slot has been reused -- in the
        // third stream
        auto pos_of = [&](int u) -> int32_t {
            if (bc.d[u] < 0) return (int32_t)((uint32_t)bc.d[u] & 0x7fffffffu);
            return fpos[((size_t)(cbase + ch) * H + u) * 64 + lane];
        };
#pragma unroll
        for (int u = 0; u < H; ++u) {
            const int32_t d = hc.p >= 0 ? bc.d[u] : 0;
            const bool near = d > 0, vec = d < 0;
            word[u] = near ? ((uint32_t)(d - 1) & (TRF_RING - 1)) * 2 : vec ? PRIV + (threadIdx.x * 8 + u) * 2 : CONSTP;
            etag[u] = near ? (uint32_t)d : vec ? TAG_PRIV : TAG_CONST;
            if (vec) vecm |= 1u << u;
            if (near) gpos = d > gpos ? d : gpos;
        }
        // Gathers from the vector.  They are requested ONE CHUNK AHEAD (the next chunk's descriptors are here by now; an operand that far back has
        // long been solved -- if not, the pair stays invalid and the general loop polls the vector), by every lane for every operand (lanes without
        // such an operand at one common address) and in front of the look-ahead, so that every wait is a COUNTED one.  (Waited for in per-operand
        // branches at the chunk's start they were two or three L2 round trips one after the other: 30 -> 23 ms; a chunk ahead: no wait at all.)
        if (__any(vecm != 0)) {
#pragma unroll
            for (int u = 0; u < H; ++u)
                if ((vecm >> u) & 1u) { const bool ok = solved(xg[u]); put_pair(word[u], xg[u], ok ? TAG_PRIV : 0u); if (!ok) vpend |= 1u << u; }
        }
        asm volatile("" ::: "memory");
#pragma unroll
        for (int u = 0; u < H; ++u)
            xg[u] = __hip_atomic_load(&out[nx.d[u] < 0 ? (int32_t)((uint32_t)nx.d[u] & 0x7fffffffu) : 0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("" ::: "memory");
        stage_of(ch + 2 * W, st);                                          // the look-ahead: in flight while this chunk and the next wait and compute
        hnn = head_of(ch + 2 * W);
        asm volatile("" ::: "memory");
        bool done = hc.p < 0, wait = !done && gpos > 0, headdone = false;
        double s = hc.s;
        int cons = 0, fails = 0;
        const uint32_t gword = ((uint32_t)(gpos - 1) & (TRF_RING - 1)) * 2, ge = (uint32_t)gpos;
        int budget = 1 << 22, since = 0, idle = 0;
        // FAST: no vector-memory load inside the loop -- the common case.  (With one anywhere in the loop body the compiler has to wait for ALL
        // outstanding loads, the look-ahead included, at the top of every iteration: the loaded registers may be the ones a poll writes.)
        // A lane that needs the vector or has a row longer than H raises `rare`, and the chunk is finished by the general form of the loop.
        auto run = [&](auto fast_tag) -> bool {
            constexpr bool FAST = decltype(fast_tag)::value;
            for (;;) {
                bool moved = false, rare = false;
                if (wait) {                                                // the gate: one word pair (a LATER tag opens it too: the pass sorts that out)
                    unsigned long long a, b; pair_at(gword, a, b);
                    const uint32_t ta = (uint32_t)(a >> 32), tb = (uint32_t)(b >> 32);
                    if ((ta < tb ? ta : tb) >= ge) wait = false;
                }
                const bool open = !wait && !done;
                ++since;
                if (__any(open) && (!__any(wait) || since >= batch)) {
                    since = 0;
                    if (open && !headdone) {
                        unsigned long long ra[H], rb[H];
#pragma unroll
                        for (int u = 0; u < H; ++u) pair_at(word[u], ra[u], rb[u]);
                        uint32_t bad = 0;                                  // (no branch per operand)
#pragma unroll
                        for (int u = 0; u < H; ++u) bad |= ((uint32_t)(ra[u] >> 32) ^ etag[u]) | ((uint32_t)(rb[u] >> 32) ^ etag[u]);
                        if (bad == 0) {
#pragma unroll
                            for (int u = 0; u < H; ++u) {                  // stored order (absent: s - 0.0 * 0.0); operand 0 of a chain's later rows IS s
                                const double x = __longlong_as_double((long long)((ra[u] & 0xffffffffull) | (rb[u] << 32)));
                                const double t = bc.v[u] * x;
                                s = (u == 0 && !hc.first) ? x : s - t;
                            }
                            headdone = true; cons = nh;
                        } else if (++fails >= 3) {
                            // rare: a ring slot that belongs to a later position now, or an operand that was not in the vector when gathered
                            if constexpr (FAST) rare = true;
                            else {
#pragma unroll
                                for (int u = 0; u < H; ++u) {
                                    const bool over = etag[u] < TAG_PRIV && ((uint32_t)(ra[u] >> 32) > etag[u] || (uint32_t)(rb[u] >> 32) > etag[u]);
                                    if (over) { word[u] = PRIV + (threadIdx.x * 8 + u) * 2; etag[u] = TAG_PRIV; put_pair(word[u], 0.0, 0u); vecm |= 1u << u; vpend |= 1u << u; }
                                    if ((vpend >> u) & 1u) {
                                        const double x = __hip_atomic_load(&out[pos_of(u)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                        if (solved(x)) { put_pair(word[u], x, TAG_PRIV); vpend &= ~(1u << u); }
                                    }
                                }
                            }
                        }
                        moved = true;
                    }
                    if (open && headdone && cons >= len) {
                        publish(hc, (FORWARD || !hc.last) ? s : s / hc.dg);
                        done = true; moved = true;
                    }
                }
                if (__all(done)) return true;
                if (FAST && __any(rare)) return false;
                if (--budget <= 0) {                                       // a logic error: NaNs, not a hung GPU
                    if (!done) publish(hc, __longlong_as_double(0x7FF8000000000000ll));
                    return true;
                }
                // a wave that makes progress is at the front: it gets the SIMD ahead of the waves that only look; those back off
                if (__any(moved)) { if (idle && (tune & 256)) __builtin_amdgcn_s_setprio(3); idle = 0; }
                else {
                    if (!idle && (tune & 256)) __builtin_amdgcn_s_setprio(0);
                    if (idle < ((tune >> 9) & 15)) ++idle;
                    for (int q = 0; q < idle; ++q) __builtin_amdgcn_s_sleep(1);
                }
            }
        };
        // LEAN: the loop a chunk normally lives in -- every unsolved lane whose gate is open looks at all its operands every turn (no batching, no
        // bookkeeping per lane: a lone wave issues an instruction every ~4.5 cycles, so the turn's LENGTH is the hop's latency).  It leaves to the
        // loops above when a lane has a row longer than H or an operand that was not in the vector, or when a ring slot has been
        // reused under a reader (looked for every 16th turn without progress).
        auto lean = [&]() -> bool {
            if (tune & 8192) return false;
            if (__any(!done && vpend != 0)) return false;
            bool stuck = false;
            for (int quiet = 0;;) {
                bool got = false;
                if (wait && (tune & 16384)) wait = false;
                if (wait) {                                                // the gate: one word pair (waves that are early keep the LDS free)
                    unsigned long long a, b; pair_at(gword, a, b);
                    const uint32_t ta = (uint32_t)(a >> 32), tb = (uint32_t)(b >> 32);
                    if ((ta < tb ? ta : tb) >= ge) wait = false;
                }
                if (!done && !wait) {
                    unsigned long long ra[H], rb[H];
#pragma unroll
                    for (int u = 0; u < H; ++u) pair_at(word[u], ra[u], rb[u]);
                    uint32_t bad = 0;
#pragma unroll
                    for (int u = 0; u < H; ++u) bad |= ((uint32_t)(ra[u] >> 32) ^ etag[u]) | ((uint32_t)(rb[u] >> 32) ^ etag[u]);
                    if (bad == 0) {
#pragma unroll
                        for (int u = 0; u < H; ++u) {                      // stored order (absent: s - 0.0 * 0.0); operand 0 of a chain's later rows IS s
                            const double x = __longlong_as_double((long long)((ra[u] & 0xffffffffull) | (rb[u] << 32)));
                            const double t = bc.v[u] * x;
                            s = (u == 0 && !hc.first) ? x : s - t;
                        }
                        publish(hc, (FORWARD || !hc.last) ? s : s / hc.dg);
                        done = true; got = true;
                    } else if ((quiet & 15) == 15) {                       // now and then: has one of my ring slots been reused under me?
#pragma unroll
                        for (int u = 0; u < H; ++u)
                            stuck = stuck || (etag[u] < TAG_PRIV && ((uint32_t)(ra[u] >> 32) > etag[u] || (uint32_t)(rb[u] >> 32) > etag[u]));
                    }
                }
                if (__all(done)) return true;
                if (__any(got)) quiet = 0;
                else { if (++quiet > 4096 || __any(stuck)) return false; __builtin_amdgcn_s_sleep(1); }
            }
        };
        if (!lean()) if (!run(std::true_type{})) run(std::false_type{});
    };
    Head hc = head_of(w), hn = head_of(w + W), hnn;
    Stage sa, sb, sc;
    double xg[H];
    stage_of(w, sa);
#pragma unroll
    for (int u = 0; u < H; ++u)
        xg[u] = __hip_atomic_load(&out[sa.d[u] < 0 ? (int32_t)((uint32_t)sa.d[u] & 0x7fffffffu) : 0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    stage_of(w + W, sb);
    for (int32_t ch = w; ch < nchunk; ch += 3 * W) {
        chunk(hc, sa, sb, sc, hnn, ch, xg);
        if (ch + W >= nchunk) break;
        hc = hn; hn = hnn;
        chunk(hc, sb, sc, sa, hnn, ch + W, xg);
        if (ch + 2 * W >= nchunk) break;
        hc = hn; hn = hnn;
        chunk(hc, sc, sa, sb, hnn, ch + 2 * W, xg);
        hc = hn; hn = hnn;
    }
    if (lane == 0) __hip_atomic_store(&wave_at[w], INT32_MAX, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// the operand streams of one narrow run (see TriFactor, tri_run_free_kernel): one thread per row of the run; `vstart` = first virtual row of
// every row of the run (and the run's total behind the last).  The streams were zeroed before: 0 = no row / no such entry.
__global__ __launch_bounds__(256) void trf_stream_kernel(const int32_t* __restrict__ ptr, const int32_t* __restrict__ col, const double* __restrict__ val,
                                                         int32_t P0, int32_t P1, int32_t reach, int32_t cbase, const int32_t* __restrict__ vstart,
                                                         int32_t* vreal, int32_t* fdesc, double* fval, int32_t* fpos) {
    const int32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P1 - P0) return;
    const int32_t p = P0 + i, k0 = ptr[p], L = ptr[p + 1] - k0, v0 = vstart[i], nv = vstart[i + 1] - v0;
    for (int32_t j = 0; j < nv; ++j) {
        const int32_t vp = v0 + j;
        const size_t cl = (size_t)(cbase + (vp >> 6)) * 64 + (size_t)(vp & 63), base = (size_t)(cbase + (vp >> 6)) * 512 + (size_t)(vp & 63);
        vreal[cl] = ((p + 1) & 0x1FFFFFFF) | (j == 0 ? 1 << 30 : 0) | (j == nv - 1 ? 1 << 29 : 0);
        for (int u = 0; u < 8; ++u) {
            int32_t d = 0, c = 0; double v = 0.0;
            if (j > 0 && u == 0) { d = vp; v = 1.0; c = p; }             // the partial sum of the virtual row before (its tag: (vp - 1) + 1)
            else {
                const int32_t e = j == 0 ? u : 8 + (j - 1) * 7 + (u - 1);                // the row's entry in this slot
                if (e < L) {
                    c = col[k0 + e]; v = val[k0 + e];
                    const int32_t vl = c >= P0 ? vstart[c - P0 + 1] - 1 : -1;             // the LAST virtual row of the operand's row holds its value
                    d = (c >= P0 && vp - vl < reach) ? vl + 1 : (int32_t)(0x80000000u | (uint32_t)c);
                }
            }
            fdesc[base + 64 * (size_t)u] = d; fval[base + 64 * (size_t)u] = v; fpos[base + 64 * (size_t)u] = c;
        }
    }
}

// runs in stream order before the level kernels, so it sees the solver's `done` flag as of this apply
__global__ void tri_set_args(TriArgs* a, const double* r, double* z, const int* done) {
    a->r = r; a->z = z; a->skip = (done && *done) ? 1 : 0;
    a->epoch = (a->epoch & 0x3fffffff) + 1;         // 1, 2, ... (d_args is zeroed at setup); never 0, the flags' initial value
}

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

// KRYST_ILU_RUN_FREE (default 1): runs of narrow levels without barriers; needs 128 KiB of LDS, twice what a kernel gets without asking
// `long_rows`: the share of the factor's rows with more than eight entries.  Such a row is a chain of virtual rows in the kernel's streams (before
// that it sent its whole chunk to a loop that took its entries from memory one at a time, and factors with more than 2 % of them were better off
// with the barrier kernel: 12 entries per row of A 32.4 ms against 22.9; as chains 12.2).  KRYST_ILU_FREE_LONG_PCT (default 100) brings the limit back.
static bool grant_free_runs(double long_rows) {
    if (env_int("KRYST_ILU_RUN_FREE", 1) == 0) return false;
    if (long_rows > 0.01 * (double)env_int("KRYST_ILU_FREE_LONG_PCT", 100) && env_int("KRYST_ILU_RUN_FREE", 1) < 2) return false;
    const bool ok = hipFuncSetAttribute((const void*)tri_run_free_kernel<true, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, TRF_LDS_BYTES) == hipSuccess &&
                    hipFuncSetAttribute((const void*)tri_run_free_kernel<false, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, TRF_LDS_BYTES) == hipSuccess;
    if (!ok) (void)hipGetLastError();
    return ok;
}

static const int NARROW = 2048;     // levels with at most this many rows are folded into one-workgroup runs (CSR fallback)

// The operand streams of every narrow run of a factor that takes tri_run_free_kernel (F->free_runs), in the order enqueue_factor meets the runs.
// `hptr`: the factor's row pointers on the host.  8 KiB per chunk of 64 virtual rows.  Not enough memory, or offsets past 32 bits: the factor keeps
// the barrier kernel (free_runs = false).
static int32_t build_free_streams(TriFactor* F, hipStream_t st, const int32_t* hptr) {
    // tri_run_free_kernel packs a row's absolute position into 29 bits of `vreal` (flags at bits 29, 30) and forms byte offsets in 32 bits:
    // a factor of 2^29 rows or more -- it would fit a 288 GB card -- takes the pipe / sync-free kernels instead
    if (F->free_runs && F->npos >= (1ll << 29)) F->free_runs = false;
    if (!F->free_runs) return KRYST_OK;
    const int nl = (int)F->lvl_off.size() - 1;
    std::vector<std::pair<int32_t, int32_t>> runs;
    std::vector<size_t> voff;                              // where a run's vstart begins in `vs`
    hvec<int32_t> vs;
    size_t chunks = 0;
    for (int lv = 0; lv < nl;) {
        if (F->lvl_off[lv + 1] - F->lvl_off[lv] <= NARROW) {
            int l1 = lv + 1;
            while (l1 < nl && F->lvl_off[l1 + 1] - F->lvl_off[l1] <= NARROW) ++l1;
            const int32_t P0 = F->lvl_off[lv], P1 = F->lvl_off[l1];
            runs.emplace_back(P0, P1);
            voff.push_back(vs.size());
            int64_t v = 0;
            vs.push_back(0);
            for (int32_t p = P0; p < P1; ++p) { const int32_t L = hptr[p + 1] - hptr[p]; v += L <= 8 ? 1 : 1 + (L - 8 + 6) / 7; vs.push_back((int32_t)std::min<int64_t>(v, INT32_MAX)); }
            if (v >= (1ll << 29)) { chunks = 0; runs.clear(); break; }
            F->run_cbase.push_back((int32_t)chunks);
            F->run_nvirt.push_back((int32_t)v);
            chunks += (size_t)(v + 63) / 64;
            lv = l1;
        } else ++lv;
    }
    int32_t* d_vs = nullptr;
    const bool ok = chunks > 0 && chunks * 512 * 8 < ((size_t)1 << 32) &&
                    pool_malloc(&F->d_fdesc, chunks * 512 * sizeof(int32_t)) == hipSuccess && pool_malloc(&F->d_fval, chunks * 512 * sizeof(double)) == hipSuccess &&
                    pool_malloc(&F->d_fpos, chunks * 512 * sizeof(int32_t)) == hipSuccess && pool_malloc(&F->d_vreal, chunks * 64 * sizeof(int32_t)) == hipSuccess &&
                    pool_malloc(&d_vs, vs.size() * sizeof(int32_t)) == hipSuccess;
    if (!ok) {
        (void)hipGetLastError();
        (void)pool_free(F->d_fdesc); (void)pool_free(F->d_fval); (void)pool_free(F->d_fpos); (void)pool_free(F->d_vreal); (void)pool_free(d_vs);
        F->d_fdesc = nullptr; F->d_fval = nullptr; F->d_fpos = nullptr; F->d_vreal = nullptr; F->run_cbase.clear(); F->run_nvirt.clear();
        F->free_runs = false;
        if (F->ell) F->syncfree = true;                    // (an ELL factor's other form)
        return KRYST_OK;
    }
    // an operand counts as "in the ring" when no wave can have reused its slot yet in the ordinary course of things: the most advanced wave is
    // held within TRF_AHEAD chunks of the least advanced one (tri_run_free_kernel: wave_at); 16 instead of 8 with a shorter reach: 20.2 ms against 20.0
    const int32_t reach = TRF_RING - 64 * (TRF_AHEAD + 1);
    hipError_t e = hipMemcpyAsync(d_vs, vs.data(), vs.size() * sizeof(int32_t), hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipMemsetAsync(F->d_fdesc, 0, chunks * 512 * sizeof(int32_t), st);
    if (e == hipSuccess) e = hipMemsetAsync(F->d_fval, 0, chunks * 512 * sizeof(double), st);
    if (e == hipSuccess) e = hipMemsetAsync(F->d_fpos, 0, chunks * 512 * sizeof(int32_t), st);
    if (e == hipSuccess) e = hipMemsetAsync(F->d_vreal, 0, chunks * 64 * sizeof(int32_t), st);
    for (size_t r = 0; e == hipSuccess && r < runs.size(); ++r) {
        const int32_t rows = runs[r].second - runs[r].first;
        hipLaunchKernelGGL(trf_stream_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, st, F->d_ptr, F->d_col, F->d_val, runs[r].first, runs[r].second,
                           reach, F->run_cbase[r], d_vs + voff[r], F->d_vreal, F->d_fdesc, F->d_fval, F->d_fpos);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    (void)pool_free(d_vs);
    if (e != hipSuccess) { set_error("operand streams of the run kernel: %s", hipGetErrorString(e)); return KRYST_ERR_HIP; }
    return KRYST_OK;
}

template <bool FORWARD>
static int32_t enqueue_factor(hipStream_t s, const TriFactor& F, const TriArgs* d_args, const double* in, double* out) {
    if (F.syncfree && F.npos > 0) {
        const dim3 grid((unsigned)((F.npos + 255) / 256));
        if (F.ell) hipLaunchKernelGGL((tri_syncfree_ell_kernel<FORWARD>), grid, dim3(256), 0, s, d_args, in, out, F.view(), F.d_diag, (int32_t)F.npos);
        // (Ilup(1) on a 7-point operator, 6 entries per row: 7.3 -> 5.9 ms per apply with 8 held, 6.3 with 16; a 27-point factor, 13 per
        // row: 9.8 -> 9.4 with 8, 6.9 with 16)
        else if (env_int("KRYST_ILU_CSR_HELD", F.held) <= 8) hipLaunchKernelGGL((tri_syncfree_csr_kernel<FORWARD, 8>), grid, dim3(256), 0, s, d_args, in, out, F.d_ptr, F.d_col, F.d_val, F.d_diag,
                                (int32_t)F.npos);
        else hipLaunchKernelGGL((tri_syncfree_csr_kernel<FORWARD, 16>), grid, dim3(256), 0, s, d_args, in, out, F.d_ptr, F.d_col, F.d_val, F.d_diag,
                                (int32_t)F.npos);
        KR_HIP(hipGetLastError());
        return KRYST_OK;
    }
    const int nl = (int)F.lvl_off.size() - 1;
    int lv = 0;
    size_t run_index = 0;
    while (lv < nl) {
        const int rows = F.lvl_off[lv + 1] - F.lvl_off[lv];
        if (rows <= NARROW && (!F.ell || F.free_runs)) {
            int l1 = lv + 1;
            while (l1 < nl && F.lvl_off[l1 + 1] - F.lvl_off[l1] <= NARROW) ++l1;
            // Threads: what a level costs is the instructions every wave of the workgroup issues for it, needed or not -- the random band
            // matrix of tools/band_apply.py (187 rows per level): 46.1 ms per apply with 1 024 threads, 39.2 with 512, 34.4 with 256, 35.8 with 320,
            // 79.5 with 128 (only a thread's first row of a level is pipelined).  Hence 5/4 of the run's mean level width, 256 at least.
            // (Not the branches around its predicated look-ahead loads: with every load unconditional -- clamped indices, straight-line code,
            // half the instructions -- the same apply took 38.7 ms, and 38.2 with the last 8 192 solution values in an LDS ring:
            // profiles/r04/narrow_level_kernel_experiments.txt.)
            const int mean_rows = (int)((F.lvl_off[l1] - F.lvl_off[lv]) / std::max(1, l1 - lv));
            const unsigned run_threads = (unsigned)std::min(1024, std::max(64, env_int("KRYST_ILU_RUN_THREADS", std::max(256, (mean_rows * 5 / 4 + 63) / 64 * 64)) / 64 * 64));
            if (F.free_runs) {
                // (waves: a power of two, so that the positions that share a ring slot belong to ONE wave and are solved in order)
                // (8 waves where a level fills a wave; 4 on narrower levels, where the waves that only wait cost the one that works:
                // 22 rows per level 0.50 us per level against 0.53, 3 rows 0.42 against 0.44; 187 rows: 30.4 ms with 8, 42.6 with 4)
                const int fw = env_int("KRYST_ILU_FREE_WAVES", mean_rows >= 64 ? 8 : 4);
                const unsigned waves = fw >= 8 ? 8u : fw >= 4 ? 4u : fw >= 2 ? 2u : 1u;
                hipLaunchKernelGGL((tri_run_free_kernel<FORWARD, 8>), dim3(1), dim3(64 * waves), (size_t)TRF_LDS_BYTES, s, d_args, in, out, F.d_diag, F.d_vreal,
                                   F.d_fdesc, F.d_fval, F.d_fpos, F.run_cbase[run_index], F.run_nvirt[run_index], env_int("KRYST_ILU_FREE_TUNE", 1 | (1 << 9) | 16384));
                ++run_index;
            } else if (env_int("KRYST_ILU_RUN_PIPE", 1) != 0)
                hipLaunchKernelGGL((tri_run_pipe_kernel<FORWARD, 8>), dim3(1), dim3(run_threads), 0, s, d_args, in, out, F.d_ptr, F.d_col, F.d_val,
                                   F.d_diag, F.d_lvl_off, lv, l1);
            else
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

// does a box-stencil factor take the pipelined wavefront kernels (KRYST_ILU_BOX: 0 no box form at all, 1 hyperplane launches, 2 wavefront)?
static bool box_takes_wavefront(const IluData* D) {
    return D->BL.ok && D->BU.ok && D->box_wave_ready && !D->safe && env_int("KRYST_ILU_PLANES", 0) == 0 && env_int("KRYST_ILU_BOX", KR_ILU_BOX_DEFAULT) >= 2;
}

// r -> rL (L level order) -> forward -> yL -> yU (U level order) -> backward -> zU -> z
static int32_t enqueue_apply(hipStream_t s, IluData* D, const TriDirect* direct = nullptr) {
    const unsigned g = (unsigned)((D->n + 255) / 256);
    if (D->BL.ok && D->BU.ok) {
        // box stencil: r -> forward -> y (natural order) -> backward -> z
        const BoxFactor& A = D->BL; const BoxFactor& B = D->BU;
        const BoxView VA{A.Ni, A.Nj, A.Nk, D->n, A.d_c, nullptr, box_stream_stride(D->n), A.d_cb}, VB{B.Ni, B.Nj, B.Nk, D->n, B.d_c, B.d_diag, box_stream_stride(D->n), B.d_cb};
        if (box_takes_wavefront(D)) {
            // pipelined wavefront over parallelograms of 8 x 8 lines (tri_box.h); the abort word behind the 2 nb flags stays set once raised
            const unsigned nb = (unsigned)(tb_nbj(A.Nj) * tb_nbk(A.Nk));
            const int budget = std::max(1, env_int("KRYST_ILU_POLL_BUDGET", 1 << 22));
            hipLaunchKernelGGL((tri_box_fill_kernel<true>), dim3(nb), dim3(256), 0, s, D->d_args, D->d_y, VA, D->d_flags, (int32_t)(2 * nb));
            const bool reg = A.regular && B.regular && env_int("KRYST_ILU_BOX_REGULAR", 1) != 0;
            // streams known at compile time: all 13 (27-point), a 19-point stencil's (lower 0x1eba / upper 0x0baf), Ilup(1)'s on a 7-point operator
            // (0x1cb0 / 0x01a7); any other set: run-time tests
            auto launch = [&](auto fwd, const BoxView& V, const double* in, double* out, int32_t* fl, uint32_t present) {
                constexpr bool F = decltype(fwd)::value;
                constexpr uint32_t FILL1 = F ? 0x1cb0u : 0x01a7u, P19 = F ? 0x1ebau : 0x0bafu;
                const size_t lds = tb_lds_bytes<F>();
#define KR_BOX_LAUNCH(R, M) hipLaunchKernelGGL((tri_box_kernel<F, R, M>), dim3(nb), dim3(256), lds, s, D->d_args, in, out, V, fl, D->d_flags + 2 * nb, D->d_gave_up, budget, present)
                if (present == 0x1fffu) { if (reg) KR_BOX_LAUNCH(true, 0x1fffu); else KR_BOX_LAUNCH(false, 0x1fffu); }
                else if (present == FILL1) { if (reg) KR_BOX_LAUNCH(true, FILL1); else KR_BOX_LAUNCH(false, FILL1); }
                else if (present == P19) { if (reg) KR_BOX_LAUNCH(true, P19); else KR_BOX_LAUNCH(false, P19); }
                else { if (reg) KR_BOX_LAUNCH(true, 0u); else KR_BOX_LAUNCH(false, 0u); }
#undef KR_BOX_LAUNCH
            };
            launch(std::true_type(), VA, (const double*)nullptr, D->d_y, D->d_flags, A.present);
            hipLaunchKernelGGL((tri_box_fill_kernel<false>), dim3(nb), dim3(256), 0, s, D->d_args, (double*)nullptr, VB, (int32_t*)nullptr, 0);
            launch(std::false_type(), VB, (const double*)D->d_y, (double*)nullptr, D->d_flags + nb, B.present);
            KR_HIP(hipGetLastError());
            return KRYST_OK;
        }
        // one launch per hyperplane i + 2 j + 4 k (KRYST_ILU_BOX=1, KRYST_ILU_PLANES=1, or after a give-up)
        const int nlev = (A.Ni - 1) + 2 * (A.Nj - 1) + 4 * (A.Nk - 1) + 1;
        const unsigned pg = (unsigned)(((int64_t)A.Nj * A.Nk + 255) / 256);
        for (int lv = 0; lv < nlev; ++lv)
            hipLaunchKernelGGL((tri_box_plane_kernel<true>), dim3(pg), dim3(256), 0, s, D->d_args, (const double*)nullptr, D->d_y, VA, lv);
        for (int lv = 0; lv < nlev; ++lv)
            hipLaunchKernelGGL((tri_box_plane_kernel<false>), dim3(pg), dim3(256), 0, s, D->d_args, (const double*)D->d_y, (double*)nullptr, VB, lv);
        KR_HIP(hipGetLastError());
        return KRYST_OK;
    }
    if (D->GL.ok && D->GU.ok) {
        // structured grid: r -> forward wavefront -> y (natural order) -> backward wavefront -> z; no permutations
        const GridFactor& A = D->GL; const GridFactor& B = D->GU;
        const unsigned nb = (unsigned)(((A.Nj + 7) / 8) * ((A.Nk + 7) / 8));
        // KRYST_ILU_WAVE: 2 = 16 x 16 lines per workgroup (tri_quad.h), 1 = 8 x 8 lines (tri_wave.h), 0 = its one-wave predecessor.
        // Default by size (measured, MI355X, true ILU(0) apply, 16 x 16 vs 8 x 8): 24^3 0.088 / 0.083 ms, 32^3 0.091 / 0.098,
        // 64^3 0.154 / 0.184, 96^3 0.217 / 0.286, 128^3 0.283 / 0.356, 256^3 0.65 / 1.00, 384^3 1.13 / 2.65, 512^3 1.91 / 5.7
        const int wave_on = env_int("KRYST_ILU_WAVE", default_wave_form(nb));
        const GridView VA{A.Ni, A.Nj, A.Nk, A.d_c1, A.d_c2, A.d_c3, nullptr}, VB{B.Ni, B.Nj, B.Nk, B.d_c1, B.d_c2, B.d_c3, B.d_diag};
        if (D->safe || env_int("KRYST_ILU_PLANES", 0)) {
            // the wavefront kernel gave up once on this preconditioner (or the caller asks for it): one launch per hyperplane
            const int nlev = A.Ni + A.Nj + A.Nk - 2;
            const unsigned pg = (unsigned)(((int64_t)A.Nj * A.Nk + 255) / 256);
            for (int lv = 0; lv < nlev; ++lv)
                hipLaunchKernelGGL((tri_plane_kernel<true>), dim3(pg), dim3(256), 0, s, D->d_args, (const double*)nullptr, D->d_y, VA, lv);
            for (int lv = 0; lv < nlev; ++lv)
                hipLaunchKernelGGL((tri_plane_kernel<false>), dim3(pg), dim3(256), 0, s, D->d_args, (const double*)D->d_y, (double*)nullptr, VB, lv);
            KR_HIP(hipGetLastError());
            return KRYST_OK;
        }
        const int budget = std::max(1, env_int("KRYST_ILU_POLL_BUDGET", 1 << 22));   // (tests force the give-up path with a tiny budget)
        if (wave_on >= 2 && A.d_blocked && B.d_blocked) {
            // 16 x 16 lines per workgroup (tri_quad.h): blocked coefficients, edge buffers between workgroups
            const unsigned nq = (unsigned)(A.nbj * A.nbk);
            const QuadView QA{A.Ni, A.Nj, A.Nk, A.nbj, A.nbk, A.nch, (const tw_v2*)A.d_blocked, A.d_edge_e, A.d_edge_n, A.d_skip};
            const QuadView QB{B.Ni, B.Nj, B.Nk, B.nbj, B.nbk, B.nch, (const tw_v2*)B.d_blocked, B.d_edge_e, B.d_edge_n, B.d_skip};
            // (no per-apply launch re-arms the edge buffers or the flags: tri_quad.h, poller / epoch)
            const TriDirect dir = direct ? *direct : TriDirect{nullptr, nullptr, nullptr, 0};
            hipLaunchKernelGGL((tri_quad_kernel<true>), dim3(nq), dim3(512), 0, s, D->d_args, (const double*)nullptr, D->d_y, QA, D->n, D->d_flags, D->d_flags + 2 * nb, D->d_gave_up, budget, dir);
            hipLaunchKernelGGL((tri_quad_kernel<false>), dim3(nq), dim3(512), 0, s, D->d_args, (const double*)D->d_y, (double*)nullptr, QB, D->n, D->d_flags + nb, D->d_flags + 2 * nb, D->d_gave_up, budget, dir);
            KR_HIP(hipGetLastError());
            return KRYST_OK;
        }
        if (wave_on > 0 && A.Ni >= 2) {
            // (the abort word behind the 2 nb flags stays set once raised: later applies of a solve that has given up leave at their
            // first poll instead of burning the poll budget again; the host switches to the plane kernels at its next sync)
            hipLaunchKernelGGL((tri_wave_fill_kernel<true>), dim3(nb), dim3(256), 0, s, D->d_args, D->d_y, VA, D->d_flags, (int32_t)(2 * nb));
            hipLaunchKernelGGL((tri_wave_kernel<true>), dim3(nb), dim3(192), 0, s, D->d_args, (const double*)nullptr, D->d_y, VA, D->n, D->d_flags, D->d_flags + 2 * nb, D->d_gave_up, budget);
            hipLaunchKernelGGL((tri_wave_fill_kernel<false>), dim3(nb), dim3(256), 0, s, D->d_args, (double*)nullptr, VB, (int32_t*)nullptr, 0);
            hipLaunchKernelGGL((tri_wave_kernel<false>), dim3(nb), dim3(192), 0, s, D->d_args, (const double*)D->d_y, (double*)nullptr, VB, D->n, D->d_flags + nb, D->d_flags + 2 * nb, D->d_gave_up, budget);
            KR_HIP(hipGetLastError());
            return KRYST_OK;
        }
        hipLaunchKernelGGL(tri_fill_kernel, dim3(g), dim3(256), 0, s, D->d_args, D->d_y, D->n);
        hipLaunchKernelGGL((tri_grid_kernel<true>), dim3(nb), dim3(64), 0, s, D->d_args, (const double*)nullptr, D->d_y,
                           GridView{A.Ni, A.Nj, A.Nk, A.d_c1, A.d_c2, A.d_c3, nullptr}, D->n);
        hipLaunchKernelGGL(tri_fill_kernel, dim3(g), dim3(256), 0, s, D->d_args, (double*)nullptr, D->n);
        hipLaunchKernelGGL((tri_grid_kernel<false>), dim3(nb), dim3(64), 0, s, D->d_args, (const double*)D->d_y, (double*)nullptr,
                           GridView{B.Ni, B.Nj, B.Nk, B.d_c1, B.d_c2, B.d_c3, B.d_diag}, D->n);
        KR_HIP(hipGetLastError());
        return KRYST_OK;
    }
    hipLaunchKernelGGL((perm_kernel<0>), dim3(g), dim3(256), 0, s, D->d_args, D->d_rL, (const double*)nullptr, D->L.d_row, D->n,
                       (D->L.syncfree || D->L.free_runs) ? D->d_y : (double*)nullptr);
    KR_HIP(hipGetLastError());
    KR_TRY(enqueue_factor<true>(s, D->L, D->d_args, D->d_rL, D->d_y));
    hipLaunchKernelGGL((perm_kernel<1>), dim3(g), dim3(256), 0, s, D->d_args, D->d_yU, (const double*)D->d_y, D->d_mapLU, D->n,
                       (D->U.syncfree || D->U.free_runs) ? D->d_zU : (double*)nullptr);
    KR_HIP(hipGetLastError());
    KR_TRY(enqueue_factor<false>(s, D->U, D->d_args, D->d_yU, D->d_zU));
    hipLaunchKernelGGL((perm_kernel<2>), dim3(g), dim3(256), 0, s, D->d_args, (double*)nullptr, (const double*)D->d_zU, D->U.d_row, D->n,
                       (double*)nullptr);
    KR_HIP(hipGetLastError());
    return KRYST_OK;
}

// does an apply take the 16 x 16 wavefront kernels (the same tests as in enqueue_apply)?
static bool takes_quad_form(const IluData* D) {
    if (!(D->GL.ok && D->GU.ok) || D->safe || env_int("KRYST_ILU_PLANES", 0)) return false;
    const unsigned nb = (unsigned)(((D->GL.Nj + 7) / 8) * ((D->GL.Nk + 7) / 8));
    return env_int("KRYST_ILU_WAVE", default_wave_form(nb)) >= 2 && D->GL.d_blocked && D->GU.d_blocked;
}

int32_t ilu_apply_dev(kryst_pc_t pc, const double* r, double* z, const int* done) {
    IluData* D = reinterpret_cast<IluData*>(pc->d_work);
    kryst_ctx_t ctx = pc->ctx;
    if (D->n == 0) return KRYST_OK;
    if (takes_quad_form(D) && env_int("KRYST_ILU_GRAPH", 0) == 0 && env_int("KRYST_ILU_DIRECT_ARGS", 1) != 0) {
        // launched directly: the caller's vectors, the `done` flag and the apply's number travel as kernel arguments (tri_quad.h: TriDirect)
        D->direct_epoch = (D->direct_epoch & 0x3fffffff) + 1;
        const TriDirect dir{r, z, done, (int32_t)(0x40000000u | (uint32_t)D->direct_epoch)};
        return enqueue_apply(ctx->s_main, D, &dir);
    }
    hipLaunchKernelGGL(tri_set_args, dim3(1), dim3(1), 0, ctx->s_main, D->d_args, r, z, done);
    KR_HIP(hipGetLastError());
    // a hipGraph pays where an apply is MANY launches (one per dependency level / hyperplane); the wavefront and the sync-free forms are
    // three to five launches, and launching them directly is 0.6 % (256^3) to 1.9 % (128^3) of a BiCGStab + ILU(0) iteration faster
    const bool few_launches = (!D->safe && env_int("KRYST_ILU_PLANES", 0) == 0 && ((D->GL.ok && D->GU.ok) || (D->L.syncfree && D->U.syncfree))) || box_takes_wavefront(D);
    const int use_graph = env_int("KRYST_ILU_GRAPH", few_launches ? 0 : 1);        // (read per apply: tools/solver_ab.py)
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
    if (D->exec && use_graph) { KR_HIP(hipGraphLaunch(D->exec, ctx->s_main)); return KRYST_OK; }
    return enqueue_apply(ctx->s_main, D);                             // eager fallback (same kernels)
}

// After a stream synchronisation: has the wavefront solve given up during an apply since the last check?  If so the
// preconditioner switches to the plane kernels for good (the captured graph is dropped) and the caller must repeat the work.
int32_t ilu_health(kryst_pc_t pc) {
    IluData* D = reinterpret_cast<IluData*>(pc->d_work);
    if (!D || !D->h_gave_up || *(volatile int32_t*)D->h_gave_up == 0) return KRYST_OK;
    *(volatile int32_t*)D->h_gave_up = 0;
    D->safe = true; D->fell_back = true;
    if (D->exec) { (void)hipGraphExecDestroy(D->exec); D->exec = nullptr; }
    if (D->graph) { (void)hipGraphDestroy(D->graph); D->graph = nullptr; }
    set_error("ILU apply: the wavefront triangular solve gave up waiting for a neighbour block (workgroups not dispatched in index "
              "order?); its result was discarded and this preconditioner now uses the level-per-launch plane kernels");
    return KRYST_SOLVE_ERROR;
}
bool ilu_fell_back(kryst_pc_t pc) {
    IluData* D = reinterpret_cast<IluData*>(pc->d_work);
    if (!D || !D->fell_back) return false;
    D->fell_back = false;
    return true;
}
bool ilu_is_wavefront(kryst_pc_t pc) {
    IluData* D = reinterpret_cast<IluData*>(pc->d_work);
    return D && ((D->GL.ok && D->GU.ok && !D->safe) || box_takes_wavefront(D));
}

void ilu_free(kryst_pc_t pc) {
    if (pc->kind != KR_PC_ILU || !pc->d_work) return;
    IluData* D = reinterpret_cast<IluData*>(pc->d_work);
    if (D->h_gave_up) (void)hipHostFree(D->h_gave_up);
    if (D->exec) (void)hipGraphExecDestroy(D->exec);
    if (D->graph) (void)hipGraphDestroy(D->graph);
    D->L.free_all(); D->U.free_all(); D->GL.free_all(); D->GU.free_all(); D->BL.free_all(); D->BU.free_all(); (void)pool_free(D->d_args); (void)pool_free(D->d_flags); (void)pool_free(D->d_y); (void)pool_free(D->d_rL); (void)pool_free(D->d_yU); (void)pool_free(D->d_zU); (void)pool_free(D->d_mapLU);
    delete D;
    pc->d_work = nullptr;
}

// Every device initialisation of a preconditioner is issued on the context's compute stream and waited for.  That stream is
// non-blocking, so the null stream does not order with it, and hipMemset on the null stream returns before the fill has run:
// round 2 zeroed the argument block with it, and under a time-sliced GPU the fill could land AFTER the first apply's
// tri_set_args had written r and z there -- the level kernels then read through a null pointer.
static thread_local hipStream_t tl_setup_stream = nullptr;       // set by the setup entry points (finish_ilu_pc, grid_setup_on_device)
static int32_t zero_dev(void* dst, size_t bytes, hipStream_t s) {
    KR_HIP(hipMemsetAsync(dst, 0, bytes, s));
    KR_HIP(hipStreamSynchronize(s));
    return KRYST_OK;
}
template <class T, class A>
static int32_t up(T** dst, const std::vector<T, A>& v) {
    KR_HIP(pool_malloc(dst, sizeof(T) * (v.size() + 1)));
    KR_HIP(hipMemsetAsync(*dst, 0, sizeof(T) * (v.size() + 1), tl_setup_stream));
    if (!v.empty()) KR_HIP(hipMemcpyAsync(*dst, v.data(), sizeof(T) * v.size(), hipMemcpyHostToDevice, tl_setup_stream));
    KR_HIP(hipStreamSynchronize(tl_setup_stream));
    return KRYST_OK;
}


// level order of one factor
static int32_t build_factor(int64_t n, const FlatRows& ent, const hvec<double>& diag, bool forward, TriFactor* F,
                            std::vector<int32_t>* pos_out) {
    std::vector<int32_t> lvl((size_t)n, 0);
    const int32_t nl = host_levels(n, ent.ptr.data(), ent.col.data(), forward, lvl.data());      // (host_factor.cpp)
    F->lvl_off.assign((size_t)nl + 1, 0);
    for (int64_t i = 0; i < n; ++i) F->lvl_off[lvl[i] + 1]++;
    for (int l = 0; l < nl; ++l) F->lvl_off[l + 1] += F->lvl_off[l];
    std::vector<int32_t> cursor(F->lvl_off.begin(), F->lvl_off.end() - 1), rowid((size_t)n), ptr((size_t)n + 1, 0);
    for (int64_t i = 0; i < n; ++i) rowid[cursor[lvl[i]]++] = (int32_t)i;        // ascending row inside a level
    std::vector<int32_t> pos((size_t)n);
    for (int64_t p = 0; p < n; ++p) pos[rowid[p]] = (int32_t)p;
    if (pos_out) *pos_out = pos;
    const size_t nnz = ent.col.size();
    hvec<int32_t> col(nnz); hvec<double> val(nnz), dg((size_t)n);          // (every element is written below: not value-initialised)
    int64_t maxlen = 0;
    for (int64_t p = 0; p < n; ++p) { const int64_t len = ent.len(rowid[p]); ptr[p + 1] = ptr[p] + (int32_t)len; maxlen = std::max(maxlen, len); }
    par_rows(n, [&](int64_t lo, int64_t hi) {
        for (int64_t p = lo; p < hi; ++p) {
            const int32_t i = rowid[p];
            size_t w = (size_t)ptr[p];
            for (int64_t k = ent.ptr[i]; k < ent.ptr[i + 1]; ++k) { col[w] = pos[ent.col[k]]; val[w] = ent.val[k]; ++w; }   // columns as level-order positions
            dg[p] = diag[i];
        }
    });
    F->npos = n;
    if (maxlen <= ELLW && n > 0) {
        hvec<int32_t> ecol((size_t)ELLW * n); hvec<double> eval((size_t)ELLW * n); hvec<uint8_t> elen((size_t)n);
        par_rows(n, [&](int64_t lo, int64_t hi) {
            for (int64_t p = lo; p < hi; ++p) {
                const int64_t len = ptr[p + 1] - ptr[p];
                elen[p] = (uint8_t)len;
                for (int64_t u = 0; u < ELLW; ++u) {                     // (padding slots: column 0, value 0)
                    ecol[(size_t)u * n + p] = u < len ? col[ptr[p] + u] : 0; eval[(size_t)u * n + p] = u < len ? val[ptr[p] + u] : 0.0;
                }
            }
        });
        KR_TRY(up(&F->d_ecol, ecol)); KR_TRY(up(&F->d_eval, eval)); KR_TRY(up(&F->d_elen, elen));
        F->ell = true;
    }
    // one sync-free launch per factor, except for CSR factors whose levels are narrow (a deep graph of a few hundred rows per level:
    // the one-workgroup run kernel's workgroup barrier, 3.4 us per level, beats two uncached round trips, 5.9 us -- measured on a
    // random band matrix with 10 716 levels of 187 rows: 127 -> 72 ms; a 27-point factor with 1 328 rows per level: 9.9 ms sync-free,
    // 13.0 ms otherwise).  KRYST_ILU_SYNCFREE = 0 / 1 forces either form.
    const double rows_per_level = nl > 0 ? (double)n / (double)nl : 0.0;
    // (round 4: an ELL factor with FEWER than 256 rows per level takes the one-workgroup barrier-free run kernel as well -- 0.45-1.4 us per level
    // there against the sync-free kernel's two fabric trips, 2 us, whatever the width)
    F->syncfree = env_int("KRYST_ILU_SYNCFREE", ((F->ell && (rows_per_level >= 256.0 || env_int("KRYST_ILU_RUN_FREE", 1) == 0)) || rows_per_level >= 512.0) ? 1 : 0) != 0;
    F->last_entry = (int32_t)ptr[n] - 1;
    {
        int64_t longest = 0, nlong = 0;
        for (int64_t p = 0; p < n; ++p) { longest = std::max<int64_t>(longest, ptr[p + 1] - ptr[p]); nlong += ptr[p + 1] - ptr[p] > 8; }
        F->held = longest <= 8 ? 8 : 16;
        F->free_runs = !F->syncfree && ptr[n] > 0 && grant_free_runs(n > 0 ? (double)nlong / (double)n : 0.0);
    }
    KR_TRY(up(&F->d_ptr, ptr)); KR_TRY(up(&F->d_col, col)); KR_TRY(up(&F->d_val, val)); KR_TRY(up(&F->d_row, rowid));
    KR_TRY(up(&F->d_diag, dg)); KR_TRY(up(&F->d_lvl_off, F->lvl_off));
    KR_TRY(build_free_streams(F, tl_setup_stream, ptr.data()));
    return KRYST_OK;
}

// Recognise a structured-grid factor and lay it out for tri_grid_kernel.  Not an error when it does not apply.
static int32_t build_grid(int64_t n, const FlatRows& ent, const hvec<double>& diag, bool forward, GridFactor* G) {
    if (n < 2 || n >= (1ll << 31) || env_int("KRYST_ILU_GRID", 1) == 0) return KRYST_OK;
    int64_t offs[3] = {0, 0, 0}; int no = 0;                              // distinct |col - row|, at most three
    {
        std::atomic<bool> reject{false};
        std::mutex mu;
        par_rows(n, [&](int64_t lo, int64_t hi) {
            int64_t mine[3] = {0, 0, 0}; int nm = 0;
            for (int64_t i = lo; i < hi && !reject.load(std::memory_order_relaxed); ++i)
                for (int64_t k = ent.ptr[i]; k < ent.ptr[i + 1]; ++k) {
                    const int64_t o = (int64_t)ent.col[k] - i;
                    if ((forward && o >= 0) || (!forward && o <= 0)) { reject = true; return; }
                    if (k > ent.ptr[i] && ent.col[k] <= ent.col[k - 1]) { reject = true; return; }   // the kernel subtracts in ascending column order (Ilut stores by magnitude)
                    const int64_t ao = o < 0 ? -o : o;
                    bool seen = false;
                    for (int q = 0; q < nm; ++q) seen = seen || mine[q] == ao;
                    if (!seen) { if (nm == 3) { reject = true; return; } mine[nm++] = ao; }
                }
            std::lock_guard<std::mutex> g(mu);
            for (int q = 0; q < nm; ++q) {
                bool seen = false;
                for (int r = 0; r < no; ++r) seen = seen || offs[r] == mine[q];
                if (!seen) { if (no == 3) { reject = true; return; } offs[no++] = mine[q]; }
            }
        });
        if (reject) return KRYST_OK;
    }
    std::sort(offs, offs + no);
    if (no < 2 || offs[0] != 1) return KRYST_OK;
    const int64_t s1 = offs[1], s2 = no == 3 ? offs[2] : n;
    if (s1 < 2 || s2 % s1 != 0 || n % s2 != 0 || s2 <= s1) return KRYST_OK;
    const int64_t Ni = s1, Nj = s2 / s1, Nk = n / s2;
    std::vector<double> c1((size_t)n, 0.0), c2((size_t)n, 0.0), c3((size_t)n, 0.0);
    {
        std::atomic<bool> reject{false};
        par_rows(n, [&](int64_t lo, int64_t hi) {
            for (int64_t i = lo; i < hi; ++i) {
                const int64_t ii = i % Ni, jx = (i / Ni) % Nj;
                for (int64_t k = ent.ptr[i]; k < ent.ptr[i + 1]; ++k) {
                    const int64_t ao = std::llabs((long long)ent.col[k] - (long long)i);
                    if (ao == 1) { if (forward ? ii == 0 : ii == Ni - 1) { reject = true; return; } c1[i] = ent.val[k]; }   // must stay on the line
                    else if (ao == s1) { if (forward ? jx == 0 : jx == Nj - 1) { reject = true; return; } c2[i] = ent.val[k]; }
                    else c3[i] = ent.val[k];
                }
            }
        });
        if (reject) return KRYST_OK;
    }
    KR_TRY(up(&G->d_c1, c1)); KR_TRY(up(&G->d_c2, c2)); KR_TRY(up(&G->d_c3, c3));
    if (!forward) KR_TRY(up(&G->d_diag, diag));
    G->Ni = (int32_t)Ni; G->Nj = (int32_t)Nj; G->Nk = (int32_t)Nk; G->ok = true;
    return KRYST_OK;
}

// Which box makes every offset of `offs` (col - row values, any sign) a step inside the 3 x 3 x 3 cube?  offset = di + Ni dj + Ni Nj dk with
// |d| <= 1.  Candidates for Ni (and for Ni Nj) are the offsets' magnitudes and their neighbours; the smallest consistent pair wins (any
// consistent pair gives a correct solve: the per-entry range checks of build_box make the decomposition unique for every row).
static bool box_decompose(int64_t o, int64_t s1, int64_t s2, int& dk, int& dj, int& di) {
    // o = di + s1 dj + s2 dk: take dk, dj as the nearest multiples
    const int64_t ao = o < 0 ? -o : o;
    for (int k = -1; k <= 1; ++k)
        for (int j = -1; j <= 1; ++j) {
            const int64_t rest = o - s2 * k - s1 * j;
            if (rest >= -1 && rest <= 1) { dk = k; dj = j; di = (int)rest; (void)ao; return true; }
        }
    return false;
}
static bool box_dims_from_offsets(const std::vector<int64_t>& offs, int64_t n, int64_t* Ni_out, int64_t* Nj_out) {
    std::vector<int64_t> mags;
    for (int64_t o : offs) { const int64_t m = o < 0 ? -o : o; if (m > 0) mags.push_back(m); }
    std::sort(mags.begin(), mags.end()); mags.erase(std::unique(mags.begin(), mags.end()), mags.end());
    if (mags.empty() || mags.size() > 13) return false;
    std::vector<int64_t> cand;
    for (int64_t m : mags) for (int64_t c : {m - 1, m, m + 1}) if (c >= 3 && c <= n) cand.push_back(c);      // lines of at least 3 rows: +1 and -1 steps are distinct from +-Ni -+ 1
    cand.push_back(n);
    std::sort(cand.begin(), cand.end()); cand.erase(std::unique(cand.begin(), cand.end()), cand.end());
    for (int64_t s1 : cand) {
        if (n % s1 != 0) continue;
        for (int64_t s2 : cand) {
            if (s2 <= s1 || s2 % s1 != 0 || n % s2 != 0 || s2 / s1 < 3) continue;      // (at least 3 lines per plane, or one plane: s2 == n)
            bool all = true;
            for (int64_t o : offs) { int dk, dj, di; if (o != 0 && !box_decompose(o, s1, s2, dk, dj, di)) { all = false; break; } if (o != 0 && s2 == n && dk != 0) { all = false; break; } }
            if (all) { *Ni_out = s1; *Nj_out = s2 / s1; return true; }
        }
    }
    return false;
}

// entries per stream -> which streams a factor has, and whether each of those has an entry wherever the neighbour row exists in the box
static void box_classify(const unsigned long long cnt[13], bool forward, int64_t Ni, int64_t Nj, int64_t Nk, BoxFactor* B) {
    B->present = 0; B->regular = true;
    for (int a = 0; a < 13; ++a) {
        const int code = forward ? a : a + 14, dk = code / 9 - 1, dj = (code / 3) % 3 - 1, di = code % 3 - 1;
        const int64_t expect = std::max<int64_t>(0, Ni - std::abs(di)) * std::max<int64_t>(0, Nj - std::abs(dj)) * std::max<int64_t>(0, Nk - std::abs(dk));
        if (cnt[a] != 0) B->present |= 1u << a;
        if (cnt[a] != 0 && (int64_t)cnt[a] != expect) B->regular = false;
    }
}

// one factor's kept entries (host FlatRows, uploaded) -> its 13 coefficient streams; `bad` is raised by an entry that is not a neighbour
// inside the box (it wraps around a line or plane end) or lies on the wrong side of the diagonal
__global__ __launch_bounds__(256) void box_rows_fill_kernel(const int64_t* __restrict__ ptr, const int32_t* __restrict__ col, const double* __restrict__ val, int32_t n,
                                                            int32_t Ni, int32_t Nj, int32_t Nk, int forward, double* c, int64_t cs, int32_t* bad, unsigned long long* counts) {
    __shared__ unsigned int cnt[13];
    if (threadIdx.x < 13) cnt[threadIdx.x] = 0;
    __syncthreads();
    const int32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) {
        const int32_t ii = i % Ni, jj = (i / Ni) % Nj, kk = i / (Ni * Nj);
        for (int64_t k = ptr[i]; k < ptr[i + 1]; ++k) {
            const int32_t j = col[k];
            if (j < 0 || j >= n) { *bad = 1; break; }
            const int di = j % Ni - ii, dj = (j / Ni) % Nj - jj, dk = j / (Ni * Nj) - kk;
            if (di < -1 || di > 1 || dj < -1 || dj > 1 || dk < -1 || dk > 1 || kk + dk >= Nk) { *bad = 1; break; }
            const int code = 9 * (dk + 1) + 3 * (dj + 1) + (di + 1);
            if (forward ? code >= 13 : code <= 13) { *bad = 1; break; }
            c[(int64_t)(forward ? code : code - 14) * cs + i] = val[k];
            atomicAdd(&cnt[forward ? code : code - 14], 1u);
        }
    }
    __syncthreads();
    if (threadIdx.x < 13 && cnt[threadIdx.x]) atomicAdd(&counts[threadIdx.x], (unsigned long long)cnt[threadIdx.x]);
}

// Recognise a box-stencil factor and lay it out as 13 natural-order coefficient streams.  Not an error when it does not apply.
static int32_t build_box(int64_t n, const FlatRows& ent, const hvec<double>& diag, bool forward, BoxFactor* B) {
    if (n < 27 || n >= (1ll << 31) || env_int("KRYST_ILU_BOX", KR_ILU_BOX_DEFAULT) == 0) return KRYST_OK;
    // distinct offsets (at most 13), rows in ascending column order, strictly lower / upper
    std::vector<int64_t> offs;
    {
        std::atomic<bool> reject{false};
        std::mutex mu;
        par_rows(n, [&](int64_t lo, int64_t hi) {
            std::vector<int64_t> mine;
            for (int64_t i = lo; i < hi && !reject.load(std::memory_order_relaxed); ++i)
                for (int64_t k = ent.ptr[i]; k < ent.ptr[i + 1]; ++k) {
                    const int64_t o = (int64_t)ent.col[k] - i;
                    if ((forward && o >= 0) || (!forward && o <= 0)) { reject = true; return; }
                    if (k > ent.ptr[i] && ent.col[k] <= ent.col[k - 1]) { reject = true; return; }   // the kernels subtract in ascending column order
                    if (std::find(mine.begin(), mine.end(), o) == mine.end()) { if (mine.size() == 13) { reject = true; return; } mine.push_back(o); }
                }
            std::lock_guard<std::mutex> g(mu);
            for (int64_t o : mine) if (std::find(offs.begin(), offs.end(), o) == offs.end()) offs.push_back(o);
        });
        if (reject || offs.empty() || offs.size() > 13) return KRYST_OK;
    }
    int64_t Ni = 0, Nj = 0;
    if (!box_dims_from_offsets(offs, n, &Ni, &Nj)) return KRYST_OK;
    const int64_t Nk = n / (Ni * Nj);
    // the streams are written on the device from the uploaded rows (round 4: the host used to fill 13 n doubles per factor and send those)
    hipStream_t st = tl_setup_stream;
    int64_t* d_ptr = nullptr; int32_t* d_col = nullptr; double* d_val = nullptr; int32_t* d_bad = nullptr;
    const size_t cb = sizeof(double) * (size_t)13 * (size_t)box_stream_stride(n), ne = (size_t)ent.ptr[(size_t)n];
    int32_t bad = 1;
    bool ok = pool_malloc(&d_ptr, sizeof(int64_t) * ((size_t)n + 1)) == hipSuccess && pool_malloc(&d_col, sizeof(int32_t) * (ne + 1)) == hipSuccess &&
              pool_malloc(&d_val, sizeof(double) * (ne + 1)) == hipSuccess && pool_malloc(&d_bad, 128) == hipSuccess && pool_malloc(&B->d_c, cb) == hipSuccess;
    unsigned long long hostw[16];                                           // [0]: "not a box factor", [1..13]: entries per stream
    ok = ok && hipMemsetAsync(B->d_c, 0, cb, st) == hipSuccess && hipMemsetAsync(d_bad, 0, 128, st) == hipSuccess &&
         hipMemcpyAsync(d_ptr, ent.ptr.data(), sizeof(int64_t) * ((size_t)n + 1), hipMemcpyHostToDevice, st) == hipSuccess &&
         hipMemcpyAsync(d_col, ent.col.data(), sizeof(int32_t) * ne, hipMemcpyHostToDevice, st) == hipSuccess &&
         hipMemcpyAsync(d_val, ent.val.data(), sizeof(double) * ne, hipMemcpyHostToDevice, st) == hipSuccess;
    if (ok) {
        hipLaunchKernelGGL(box_rows_fill_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, d_ptr, d_col, d_val, (int32_t)n, (int32_t)Ni, (int32_t)Nj, (int32_t)Nk,
                           forward ? 1 : 0, B->d_c, box_stream_stride(n), d_bad, reinterpret_cast<unsigned long long*>(d_bad) + 1);
        ok = hipGetLastError() == hipSuccess && hipMemcpyAsync(hostw, d_bad, 128, hipMemcpyDeviceToHost, st) == hipSuccess && hipStreamSynchronize(st) == hipSuccess;
        if (ok) bad = (int32_t)(hostw[0] & 0xffffffffull);
    }
    (void)pool_free(d_ptr); (void)pool_free(d_col); (void)pool_free(d_val); (void)pool_free(d_bad);
    if (!ok) { (void)hipGetLastError(); B->free_all(); set_error("box-stencil factor: device allocation or copy failed"); return KRYST_ERR_HIP; }
    if (bad != 0) { B->free_all(); return KRYST_OK; }
    if (!forward) KR_TRY(up(&B->d_diag, diag));
    box_classify(hostw + 1, forward, Ni, Nj, Nk, B);
    B->Ni = (int32_t)Ni; B->Nj = (int32_t)Nj; B->Nk = (int32_t)Nk; B->ok = true;
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
// device buffers every ILU-family preconditioner needs once its factors are on the device (grid form or level-ordered form)
static int32_t finish_ilu_device(kryst_pc_t pc, IluData* D) {
    kryst_ctx_t ctx = pc->ctx;
    const int64_t n = D->n;
    int32_t rc = KRYST_OK;
    if (pool_malloc(&D->d_args, sizeof(TriArgs)) != hipSuccess || zero_dev(D->d_args, sizeof(TriArgs), ctx->s_main) != KRYST_OK) { set_error("hipMalloc failed"); rc = KRYST_ERR_HIP; }
    if (rc == KRYST_OK) {
        const size_t bytes = sizeof(double) * (size_t)((n + KR_TILE - 1) / KR_TILE * KR_TILE + KR_TILE);
        const bool grid = (D->GL.ok && D->GU.ok) || (D->BL.ok && D->BU.ok);   // the wavefront / box solves work in place: one intermediate vector
        for (double** pp : {&D->d_y, &D->d_rL, &D->d_yU, &D->d_zU}) {
            if (rc != KRYST_OK) break;
            if (grid && pp != &D->d_y) continue;
            if (pool_malloc(pp, bytes) != hipSuccess) { set_error("hipMalloc failed"); rc = KRYST_ERR_HIP; }
            else if (hipMemsetAsync(*pp, 0, bytes, ctx->s_main) != hipSuccess) rc = KRYST_ERR_HIP;
        }
        if (rc == KRYST_OK && hipStreamSynchronize(ctx->s_main) != hipSuccess) rc = KRYST_ERR_HIP;
    }
    if (rc == KRYST_OK && D->BL.ok && D->BU.ok && D->BL.Ni >= 2) {
        const size_t nb = (size_t)tb_nbj(D->BL.Nj) * (size_t)tb_nbk(D->BL.Nk);
        if (pool_malloc(&D->d_flags, sizeof(int32_t) * (2 * nb + 1)) != hipSuccess || zero_dev(D->d_flags, sizeof(int32_t) * (2 * nb + 1), ctx->s_main) != KRYST_OK) { set_error("hipMalloc failed"); rc = KRYST_ERR_HIP; }
        if (rc == KRYST_OK && (hipHostMalloc((void**)&D->h_gave_up, 64, hipHostMallocMapped) != hipSuccess ||
                               hipHostGetDevicePointer((void**)&D->d_gave_up, D->h_gave_up, 0) != hipSuccess)) { set_error("hipHostMalloc failed"); rc = KRYST_ERR_HIP; }
        if (rc == KRYST_OK && env_int("KRYST_ILU_BOX", KR_ILU_BOX_DEFAULT) >= 2) {
            // blocked copies of both factors for the wavefront kernels' loaders, one device pass each
            const size_t nch = (size_t)(D->BL.Ni + 29 + TB_C - 1) / TB_C;
            const size_t el = (size_t)(TB_C / 2) * 64 * 16;
            const BoxView VA{D->BL.Ni, D->BL.Nj, D->BL.Nk, D->n, D->BL.d_c, nullptr, box_stream_stride(D->n), nullptr};
            const BoxView VB{D->BU.Ni, D->BU.Nj, D->BU.Nk, D->n, D->BU.d_c, D->BU.d_diag, box_stream_stride(D->n), nullptr};
            if (pool_malloc(&D->BL.d_cb, nb * nch * 13 * el) != hipSuccess || pool_malloc(&D->BU.d_cb, nb * nch * 14 * el) != hipSuccess) {
                (void)hipGetLastError(); (void)pool_free(D->BL.d_cb); (void)pool_free(D->BU.d_cb); D->BL.d_cb = D->BU.d_cb = nullptr;     // (no room: the hyperplane kernels)
            } else {
                hipLaunchKernelGGL((tri_box_layout_kernel<true>), dim3((unsigned)(nb * nch)), dim3(256), 0, ctx->s_main, VA, (tw_v2*)D->BL.d_cb);
                hipLaunchKernelGGL((tri_box_layout_kernel<false>), dim3((unsigned)(nb * nch)), dim3(256), 0, ctx->s_main, VB, (tw_v2*)D->BU.d_cb);
                if (hipGetLastError() != hipSuccess || hipStreamSynchronize(ctx->s_main) != hipSuccess) { set_error("box layout kernel failed"); rc = KRYST_ERR_HIP; }
            }
        }
        if (rc == KRYST_OK) {
            *D->h_gave_up = 0;
            // 70 / 73 KiB of LDS per workgroup (two per CU): more than the 64 KiB a kernel gets without asking
            bool ok = true;
#define KR_BOX_ATTR(F, R, M) ok = ok && hipFuncSetAttribute((const void*)tri_box_kernel<F, R, M>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)tb_lds_bytes<F>()) == hipSuccess
            KR_BOX_ATTR(true, true, 0x1ebau); KR_BOX_ATTR(true, false, 0x1ebau); KR_BOX_ATTR(false, true, 0x0bafu); KR_BOX_ATTR(false, false, 0x0bafu);
            KR_BOX_ATTR(true, true, 0x1fffu); KR_BOX_ATTR(true, true, 0x1cb0u); KR_BOX_ATTR(true, true, 0u); KR_BOX_ATTR(true, false, 0x1fffu); KR_BOX_ATTR(true, false, 0x1cb0u); KR_BOX_ATTR(true, false, 0u);
            KR_BOX_ATTR(false, true, 0x1fffu); KR_BOX_ATTR(false, true, 0x01a7u); KR_BOX_ATTR(false, true, 0u); KR_BOX_ATTR(false, false, 0x1fffu); KR_BOX_ATTR(false, false, 0x01a7u); KR_BOX_ATTR(false, false, 0u);
#undef KR_BOX_ATTR
            D->box_wave_ready = ok && D->BL.d_cb && D->BU.d_cb;
            (void)hipGetLastError();
        }
    }
    if (rc == KRYST_OK && D->GL.ok && D->GU.ok) {
        const size_t nb = (size_t)((D->GL.Nj + 7) / 8) * (size_t)((D->GL.Nk + 7) / 8);
        if (pool_malloc(&D->d_flags, sizeof(int32_t) * (2 * nb + 1)) != hipSuccess || zero_dev(D->d_flags, sizeof(int32_t) * (2 * nb + 1), ctx->s_main) != KRYST_OK) { set_error("hipMalloc failed"); rc = KRYST_ERR_HIP; }
        if (rc == KRYST_OK && (hipHostMalloc((void**)&D->h_gave_up, 64, hipHostMallocMapped) != hipSuccess ||
                               hipHostGetDevicePointer((void**)&D->d_gave_up, D->h_gave_up, 0) != hipSuccess)) { set_error("hipHostMalloc failed"); rc = KRYST_ERR_HIP; }
        if (rc == KRYST_OK) *D->h_gave_up = 0;
        if (rc == KRYST_OK && D->GL.Ni >= 2 && env_int("KRYST_ILU_WAVE", default_wave_form(nb)) >= 2) {
            // blocked coefficient layout + edge buffers of the 16 x 16 kernel (tri_quad.h), one device pass per factor
            for (GridFactor* G : {&D->GL, &D->GU}) {
                if (rc != KRYST_OK) break;
                const bool fwd = G == &D->GL;
                G->nbj = (G->Nj + 15) / 16; G->nbk = (G->Nk + 15) / 16; G->nch = (G->Ni + 14 + TQ_C - 1) / TQ_C;
                const size_t nq = (size_t)G->nbj * G->nbk;
                const int NA = fwd ? 3 : 4;
                const size_t cbytes = nq * G->nch * NA * 4 * TQ_LINES * sizeof(tw_v2);
                const size_t ebytes = nq * (size_t)(G->nch * TQ_C + 8) * 16 * sizeof(double);
                if (pool_malloc(&G->d_blocked, cbytes) != hipSuccess || pool_malloc(&G->d_edge_e, ebytes) != hipSuccess ||
                    pool_malloc(&G->d_edge_n, ebytes) != hipSuccess) {
                    // no room for the blocked copy (it doubles the factor's footprint): the 8 x 8 kernel works on the natural-order streams
                    (void)hipGetLastError();
                    for (GridFactor* H : {&D->GL, &D->GU}) {
                        (void)pool_free(H->d_blocked); (void)pool_free(H->d_edge_e); (void)pool_free(H->d_edge_n); (void)pool_free(H->d_skip);
                        H->d_blocked = nullptr; H->d_edge_e = H->d_edge_n = nullptr; H->d_skip = nullptr;
                    }
                    break;
                }
                const GridView V{G->Ni, G->Nj, G->Nk, G->d_c1, G->d_c2, G->d_c3, G->d_diag};
                const unsigned lg = (unsigned)(nq * G->nch);
                if (fwd) hipLaunchKernelGGL((tri_quad_layout_kernel<true, 3>), dim3(lg), dim3(256), 0, ctx->s_main, V, G->nbj, G->nbk, G->nch, (tw_v2*)G->d_blocked);
                else hipLaunchKernelGGL((tri_quad_layout_kernel<false, 4>), dim3(lg), dim3(256), 0, ctx->s_main, V, G->nbj, G->nbk, G->nch, (tw_v2*)G->d_blocked);
                // edge buffers armed once (sentinels; zeros past the last step): between applies the pollers re-arm what they consume
                hipLaunchKernelGGL(tri_quad_fill_kernel, dim3(std::min<unsigned>(1024u, (unsigned)nq * 4u)), dim3(256), 0, ctx->s_main, G->d_edge_e, G->d_edge_n, (int)nq, G->nch, (int32_t*)nullptr, 0);
                // which chunks repeat chunk - 3 bit for bit (their coefficients are in the solving wave's registers already)
                if (env_int("KRYST_ILU_DEDUP", 1) && G->nch + 3 <= TQ_SKIPMAX) {
                    if (pool_malloc(&G->d_skip, nq * 4 * (size_t)G->nch) != hipSuccess) { (void)hipGetLastError(); G->d_skip = nullptr; }      // (flags are optional)
                    else if (fwd) hipLaunchKernelGGL((tri_quad_dedup_kernel<3>), dim3(lg), dim3(256), 0, ctx->s_main, (const tw_v2*)G->d_blocked, G->nch, G->d_skip);
                    else hipLaunchKernelGGL((tri_quad_dedup_kernel<4>), dim3(lg), dim3(256), 0, ctx->s_main, (const tw_v2*)G->d_blocked, G->nch, G->d_skip);
                }
                if (hipGetLastError() != hipSuccess || hipStreamSynchronize(ctx->s_main) != hipSuccess) { set_error("blocked layout kernel failed"); rc = KRYST_ERR_HIP; }
                if (rc == KRYST_OK && G->d_skip && getenv("KRYST_ILU_VERBOSE")) {
                    std::vector<uint8_t> h(nq * 4 * (size_t)G->nch);
                    if (hipMemcpyAsync(h.data(), G->d_skip, h.size(), hipMemcpyDeviceToHost, ctx->s_main) == hipSuccess && hipStreamSynchronize(ctx->s_main) == hipSuccess) { G->nskip = 0; for (uint8_t v : h) G->nskip += v; }
                    fprintf(stderr, "[kryst ilu] %s factor: %lld of %zu coefficient chunks repeat (no request)\n", fwd ? "forward" : "backward", (long long)G->nskip, h.size());
                }
            }
        }
    }
    if (getenv("KRYST_ILU_VERBOSE")) {
        if (D->BL.ok) { fprintf(stderr, "[kryst ilu] box streams: L present %#x regular %d, U present %#x regular %d\n", D->BL.present, (int)D->BL.regular, D->BU.present, (int)D->BU.regular); }
        fprintf(stderr, "[kryst ilu] n=%lld %s; levels L/U=%zu/%zu\n", (long long)n, D->GL.ok ? "structured grid (wavefront kernel)" : D->BL.ok ? "box stencil (13 streams per factor)" : "level-ordered",
                D->L.lvl_off.empty() ? (size_t)0 : D->L.lvl_off.size() - 1, D->U.lvl_off.empty() ? (size_t)0 : D->U.lvl_off.size() - 1);
    }
    return rc;
}

// shared tail of every host-side ILU-family setup: recognise a structured grid or level-order both factors, then hand out the
// preconditioner object
static int32_t finish_ilu_pc(kryst_csr_t a, int mode, bool divide, const FlatRows& le, const FlatRows& ue,
                             const hvec<double>& dg, kryst_pc_t* out) {
    kryst_ctx_t ctx = a->ctx;
    tl_setup_stream = ctx->s_main;
    const int64_t n = a->nrows;
    hvec<double> ones((size_t)n, 1.0);
    kryst_pc_t pc = new kryst_pc_s();
    pc->ctx = ctx; pc->kind = KR_PC_ILU; pc->a = a; pc->n = n; pc->ilu_mode = mode; pc->divide_diag = divide;
    IluData* D = new IluData();
    D->n = n;
    pc->d_work = reinterpret_cast<double*>(D);
    // structured-grid factors take the wavefront kernel; everything else is level-ordered
    int32_t rc = build_grid(n, le, ones, true, &D->GL);
    if (rc == KRYST_OK) rc = build_grid(n, ue, dg, false, &D->GU);
    if (D->GL.ok && D->GU.ok && (D->GL.Ni != D->GU.Ni || D->GL.Nj != D->GU.Nj)) D->GL.ok = false;
    if (rc == KRYST_OK && !(D->GL.ok && D->GU.ok)) {
        D->GL.free_all(); D->GU.free_all(); D->GL = GridFactor(); D->GU = GridFactor();
        // box stencils wider than 7 points (round 4): natural-order streams, hyperplane / wavefront kernels, no level machinery
        rc = build_box(n, le, ones, true, &D->BL);
        if (rc == KRYST_OK && D->BL.ok) rc = build_box(n, ue, dg, false, &D->BU);
        if (!(D->BL.ok && D->BU.ok && D->BL.Ni == D->BU.Ni && D->BL.Nj == D->BU.Nj)) { D->BL.free_all(); D->BU.free_all(); D->BL = BoxFactor(); D->BU = BoxFactor(); }
    }
    if (rc == KRYST_OK && !(D->GL.ok && D->GU.ok) && !(D->BL.ok && D->BU.ok)) {
        std::vector<int32_t> posL, posU;
        // the two factors' level orders are independent: side by side (each has its own host arrays; the uploads share the context's stream)
        // (the error text is thread-local and an exception must not leave a thread function: the worker hands both back -- ADVICE r04)
        int32_t rc_u = KRYST_OK;
        std::string err_u;
        std::thread upper([&] {
            try {
                (void)hipSetDevice(ctx->device);
                tl_setup_stream = ctx->s_main;
                rc_u = build_factor(n, ue, dg, false, &D->U, &posU);
                if (rc_u != KRYST_OK) err_u = kryst_hip_last_error();
            } catch (const std::bad_alloc&) { rc_u = KRYST_ERR_ARG; err_u = "ILU set-up: out of host memory while level-ordering U";
            } catch (...) { rc_u = KRYST_ERR_ARG; err_u = "ILU set-up: unexpected exception while level-ordering U"; }
        });
        try { rc = build_factor(n, le, ones, true, &D->L, &posL); }
        catch (const std::bad_alloc&) { rc = KRYST_ERR_ARG; set_error("ILU set-up: out of host memory while level-ordering L"); }
        upper.join();
        if (rc == KRYST_OK && rc_u != KRYST_OK) { rc = rc_u; set_error("%s", err_u.c_str()); }
        if (rc == KRYST_OK) {
            std::vector<int32_t> mapLU((size_t)n);
            for (int64_t i = 0; i < n; ++i) mapLU[posU[i]] = posL[i];
            rc = up(&D->d_mapLU, mapLU);
        }
    }
    if (rc == KRYST_OK) rc = finish_ilu_device(pc, D);
    if (rc != KRYST_OK) { kryst_pc_destroy(pc); return rc; }
    *out = pc;
    return KRYST_OK;
}

// ---------------------------------------------------------------------------------------------------------------------------
// Device-side setup for grid operators.  A matrix whose stored entries couple row (i,j,k) of an Ni x Nj x Nk box only to itself
// and its six axis neighbours, in natural ordering (recognised from the CSR-D8 offset dictionary: at most the offsets 0, +-1,
// +-Ni, +-Ni*Nj, no entry wrapping around a line end), never leaves the GPU: one pass pulls the seven coefficient streams out
// of the CSR arrays, the compat / Ilup(0) factors are pointwise quotients, and the textbook ILU(0) -- on this pattern only the
// DIAGONAL is ever updated, u_dd(i) = a_dd - sum_c l_ic u_ci over the lower neighbours c in ascending column order -- is a
// recurrence over the hyperplanes i + j + k = const: one small launch per plane (766 at 256^3, ~3 ms).  Same operations in the
// same order as the host loops above (which remain for every other matrix): same bits.  256^3: 1.8 s -> a few ms.
struct GridStreams {                 // natural row order, n entries each; absent entry = 0.0 and its presence bit clear
    double *km, *jm, *im, *dd, *ip, *jp, *kp; uint8_t* have;     // presence bits: 0 km, 1 jm, 2 im, 3 dd, 4 ip, 5 jp, 6 kp
};

__global__ __launch_bounds__(256) void grid_extract_kernel(const int32_t* __restrict__ rp, const int32_t* __restrict__ col, const double* __restrict__ val,
                                                           int64_t n, int32_t s1, int64_t s2, GridStreams G, int32_t* reject) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    double v[7] = {0, 0, 0, 0, 0, 0, 0};
    unsigned have = 0;
    const int32_t ii = (int32_t)(i % s1), jx = (int32_t)((i / s1) % (s2 / s1));
    const int32_t nj = (int32_t)(s2 / s1);
    bool bad = false;
    for (int32_t k = rp[i]; k < rp[i + 1]; ++k) {
        const int64_t o = (int64_t)col[k] - i;
        int slot = -1;
        if (o == -s2) slot = 0; else if (o == -(int64_t)s1) slot = 1; else if (o == -1) slot = 2; else if (o == 0) slot = 3;
        else if (o == 1) slot = 4; else if (o == (int64_t)s1) slot = 5; else if (o == s2) slot = 6;
        if (slot < 0) { bad = true; continue; }
        // an entry must stay on its grid line / plane (a band that wraps around a line end is not a grid operator)
        if ((slot == 2 && ii == 0) || (slot == 4 && ii == s1 - 1) || (slot == 1 && jx == 0) || (slot == 5 && jx == nj - 1)) bad = true;
        v[slot] = val[k]; have |= 1u << slot;
    }
    if (bad) atomicOr(reject, 1);
    G.km[i] = v[0]; G.jm[i] = v[1]; G.im[i] = v[2]; G.dd[i] = v[3]; G.ip[i] = v[4]; G.jp[i] = v[5]; G.kp[i] = v[6];
    G.have[i] = (uint8_t)have;
}

// which of the 256 offset codes occur at all (the stencil generator's dictionary also lists halo offsets it may not use)
__global__ __launch_bounds__(256) void code_usage_kernel(const uint8_t* __restrict__ code, int64_t nnz, int32_t* used) {
    __shared__ int32_t mine[256];
    mine[threadIdx.x] = 0;
    __syncthreads();
    for (int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x; k < nnz; k += (int64_t)gridDim.x * 256) mine[code[k]] = 1;
    __syncthreads();
    if (mine[threadIdx.x]) used[threadIdx.x] = 1;
}

// Ilu0 as written (mode 0) and Ilup::new(0) (mode 1): l_ij = a_ij / a_jj for stored nonzeros below the diagonal, U = triu(A)
// (ilu.rs:76-80, ilup.rs:104-111); bad_row: lowest row whose pivot is zero (Ilup only, ilup.rs:106-108)
__global__ __launch_bounds__(256) void grid_pointwise_factor_kernel(GridStreams G, int64_t n, int32_t s1, int64_t s2, int mode,
                                                                    double* l1, double* l2, double* l3, double* u1, double* u2, double* u3, double* dg,
                                                                    long long* bad_row) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const unsigned have = G.have[i];
    const double a[3] = {G.km[i], G.jm[i], G.im[i]};
    const int64_t nb[3] = {i - s2, i - s1, i - 1};
    double l[3];
    bool zero_pivot = false;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        l[c] = 0.0;
        if (((have >> c) & 1u) && a[c] != 0.0) {
            const double ujj = (G.have[nb[c]] & 8u) ? G.dd[nb[c]] : 0.0;
            if (mode == KRYST_ILU_ILUP0 && ujj == 0.0) zero_pivot = true;
            l[c] = a[c] / ujj;
        }
    }
    if (zero_pivot) atomicMin((unsigned long long*)bad_row, (unsigned long long)i);
    auto kept = [](double x) { return x == 0.0 ? 0.0 : x; };               // a zero entry (either sign) is not kept: +0.0 = "no entry"
    l3[i] = kept(l[0]); l2[i] = kept(l[1]); l1[i] = kept(l[2]);            // coefficient of the k- / j- / i-neighbour
    u1[i] = kept(G.ip[i]); u2[i] = kept(G.jp[i]); u3[i] = kept(G.kp[i]);
    dg[i] = (mode != KRYST_ILU_KRYST_COMPAT && (have & 8u)) ? G.dd[i] : 1.0;   // ilu.rs:115-119 never divides; ilup.rs:160-164 (missing diagonal: no divide)
}

// Textbook ILU(0), the rows of ONE hyperplane (they only need lower planes): the host loop's operations in its order --
// for c ascending: pivot check, l = a_ic / u_cc, then u_ii -= l * u_ci when row c reaches i and row i has a diagonal.
__global__ __launch_bounds__(256) void grid_true_ilu0_plane_kernel(GridStreams G, int32_t Ni, int32_t Nj, int32_t Nk, int level,
                                                                   double* wdd, double* l1, double* l2, double* l3, long long* bad_key) {
    const int jk = blockIdx.x * 256 + threadIdx.x;
    if (jk >= Nj * Nk) return;
    const int j = jk % Nj, k = jk / Nj, ii = level - j - k;
    if (ii < 0 || ii >= Ni) return;
    const int64_t s1 = Ni, s2 = (int64_t)Ni * Nj, i = ii + s1 * j + s2 * k;
    const unsigned have = G.have[i];
    const double a[3] = {G.km[i], G.jm[i], G.im[i]};
    const int64_t nb[3] = {i - s2, i - s1, i - 1};
    const double* up[3] = {G.kp, G.jp, G.ip};                              // row c's entry that points at row i
    const unsigned upbit[3] = {64u, 32u, 16u};
    double wd = G.dd[i];
    double l[3] = {0.0, 0.0, 0.0};
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        if (!((have >> c) & 1u)) continue;
        const unsigned hc = G.have[nb[c]];
        const double piv = wdd[nb[c]];
        if (!(hc & 8u) || piv == 0.0) atomicMin((unsigned long long*)bad_key, (unsigned long long)(i * 4 + c));
        l[c] = a[c] / piv;
        if ((hc & upbit[c]) && (have & 8u)) wd = wd - l[c] * up[c][nb[c]];
    }
    wdd[i] = wd;
    auto kept = [](double x) { return x == 0.0 ? 0.0 : x; };
    l3[i] = kept(l[0]); l2[i] = kept(l[1]); l1[i] = kept(l[2]);
}

// -> KRYST_OK and *out set when the operator was handled on the device; *out left null when it is not a grid operator (the host
// path takes over); an error code for a zero pivot.
static int32_t grid_setup_on_device(kryst_csr_t a, int mode, kryst_pc_t* out) {
    *out = nullptr;
    kryst_ctx_t ctx = a->ctx;
    const int64_t n = a->nrows;
    if (a->dist || !a->d_dict || !a->d_code || n < 27 || n >= (1ll << 31) || env_int("KRYST_ILU_GRID", 1) == 0 || env_int("KRYST_ILU_DEVICE_SETUP", 1) == 0)
        return KRYST_OK;
    int32_t dict[256], used[256];
    {
        int32_t* d_used = nullptr;
        KR_HIP(pool_malloc(&d_used, sizeof used));
        (void)hipMemsetAsync(d_used, 0, sizeof used, ctx->s_main);
        hipLaunchKernelGGL(code_usage_kernel, dim3((unsigned)std::min<int64_t>(4096, (a->nnz + 255) / 256 + 1)), dim3(256), 0, ctx->s_main, a->d_code, a->nnz, d_used);
        const hipError_t e1 = hipMemcpyAsync(used, d_used, sizeof used, hipMemcpyDeviceToHost, ctx->s_main);
        const hipError_t e2 = hipMemcpyAsync(dict, a->d_dict, sizeof dict, hipMemcpyDeviceToHost, ctx->s_main);
        const hipError_t e3 = hipStreamSynchronize(ctx->s_main);
        (void)pool_free(d_used);
        if (e1 != hipSuccess || e2 != hipSuccess || e3 != hipSuccess) { set_error("code usage scan failed"); return KRYST_ERR_HIP; }
    }
    int64_t offs[3] = {0, 0, 0}; int no = 0;
    for (int q = 0; q < 256; ++q) {
        if (!used[q]) continue;
        const int64_t ao = dict[q] < 0 ? -(int64_t)dict[q] : dict[q];
        if (ao == 0) continue;
        bool seen = false;
        for (int r = 0; r < no; ++r) seen = seen || offs[r] == ao;
        if (!seen) { if (no == 3) return KRYST_OK; offs[no++] = ao; }
    }
    std::sort(offs, offs + no);
    if (no < 2 || offs[0] != 1) return KRYST_OK;
    const int64_t s1 = offs[1], s2 = no == 3 ? offs[2] : n;
    // Ni >= 3 and Nj >= 3 (in 3-D): with a dimension of 2 a lower neighbour's row reaches a SECOND entry of row i and the
    // elimination is no longer confined to the diagonal
    if (s1 < 3 || s2 % s1 != 0 || n % s2 != 0 || s2 <= s1 || (no == 3 && s2 / s1 < 3)) return KRYST_OK;
    const int32_t Ni = (int32_t)s1, Nj = (int32_t)(s2 / s1), Nk = (int32_t)(n / s2);
    const bool verbose = getenv("KRYST_ILU_VERBOSE") != nullptr;
    const auto t0 = std::chrono::steady_clock::now();
    // ---- seven coefficient streams + presence bits
    GridStreams G{};
    double* work[8] = {};
    struct Guard { double** w; uint8_t** h; long long** f; int32_t** r; ~Guard() { for (int i = 0; i < 8; ++i) (void)pool_free(w[i]); (void)pool_free(*h); (void)pool_free(*f); (void)pool_free(*r); } };
    uint8_t* d_have = nullptr; long long* d_flag = nullptr; int32_t* d_reject = nullptr;
    Guard guard{work, &d_have, &d_flag, &d_reject};
    const size_t vb = sizeof(double) * (size_t)(n + 8);
    for (int q = 0; q < 8; ++q) KR_HIP(pool_malloc(&work[q], vb));
    KR_HIP(pool_malloc(&d_have, (size_t)n + 8)); KR_HIP(pool_malloc(&d_flag, 16)); KR_HIP(pool_malloc(&d_reject, 4));
    KR_HIP(hipMemsetAsync(d_reject, 0, 4, ctx->s_main));
    KR_HIP(hipMemsetAsync(d_flag, 0xff, 16, ctx->s_main));
    G.km = work[0]; G.jm = work[1]; G.im = work[2]; G.dd = work[3]; G.ip = work[4]; G.jp = work[5]; G.kp = work[6]; G.have = d_have;
    const unsigned rg = (unsigned)((n + 255) / 256);
    hipLaunchKernelGGL(grid_extract_kernel, dim3(rg), dim3(256), 0, ctx->s_main, a->d_row_ptr, a->d_col, a->d_val, n, Ni, s2, G, d_reject);
    KR_HIP(hipGetLastError());
    int32_t reject = 0;
    KR_HIP(hipMemcpyAsync(&reject, d_reject, 4, hipMemcpyDeviceToHost, ctx->s_main));
    KR_HIP(hipStreamSynchronize(ctx->s_main));
    if (reject) return KRYST_OK;                                           // bands that wrap around line ends: the host path (level-ordered)
    // ---- the preconditioner object with its natural-order factor streams
    kryst_pc_t pc = new kryst_pc_s();
    const bool divide = mode != KRYST_ILU_KRYST_COMPAT;
    pc->ctx = ctx; pc->kind = KR_PC_ILU; pc->a = a; pc->n = n; pc->ilu_mode = mode; pc->divide_diag = divide;
    IluData* D = new IluData();
    D->n = n;
    pc->d_work = reinterpret_cast<double*>(D);
    int32_t rc = KRYST_OK;
    for (double** pp : {&D->GL.d_c1, &D->GL.d_c2, &D->GL.d_c3, &D->GU.d_c1, &D->GU.d_c2, &D->GU.d_c3, &D->GU.d_diag})
        if (rc == KRYST_OK && pool_malloc(pp, vb) != hipSuccess) { set_error("hipMalloc failed"); rc = KRYST_ERR_HIP; }
    if (rc == KRYST_OK) {
        if (mode == KRYST_ILU_TRUE_ILU0) {
            double* wdd = work[7];
            const unsigned pg = (unsigned)(((int64_t)Nj * Nk + 255) / 256);
            for (int lv = 0; lv < Ni + Nj + Nk - 2; ++lv)
                hipLaunchKernelGGL(grid_true_ilu0_plane_kernel, dim3(pg), dim3(256), 0, ctx->s_main, G, Ni, Nj, Nk, lv, wdd, D->GL.d_c1, D->GL.d_c2, D->GL.d_c3, d_flag);
            // U = the upper entries of A (never touched on this pattern) and, as divisor, the eliminated diagonal where the row
            // has one, else 1 (ilup.rs:160-164): the pointwise kernel with the eliminated diagonal in place of A's (mode -1: its
            // L outputs go to scratch that is no longer needed)
            hipLaunchKernelGGL(grid_pointwise_factor_kernel, dim3(rg), dim3(256), 0, ctx->s_main,
                               GridStreams{G.km, G.jm, G.im, wdd, G.ip, G.jp, G.kp, G.have}, n, Ni, s2, -1,
                               work[0], work[1], work[2], D->GU.d_c1, D->GU.d_c2, D->GU.d_c3, D->GU.d_diag, d_flag + 1);
        } else {
            hipLaunchKernelGGL(grid_pointwise_factor_kernel, dim3(rg), dim3(256), 0, ctx->s_main, G, n, Ni, s2, mode,
                               D->GL.d_c1, D->GL.d_c2, D->GL.d_c3, D->GU.d_c1, D->GU.d_c2, D->GU.d_c3, D->GU.d_diag, d_flag);
        }
        if (rc == KRYST_OK && hipGetLastError() != hipSuccess) rc = KRYST_ERR_HIP;
    }
    long long flag[2] = {-1, -1};
    if (rc == KRYST_OK && (hipMemcpyAsync(flag, d_flag, 16, hipMemcpyDeviceToHost, ctx->s_main) != hipSuccess || hipStreamSynchronize(ctx->s_main) != hipSuccess)) rc = KRYST_ERR_HIP;
    if (rc == KRYST_OK && flag[0] != -1) {
        if (mode == KRYST_ILU_TRUE_ILU0) {
            const long long i = flag[0] / 4; const int c = (int)(flag[0] % 4);
            const long long row = c == 0 ? i - s2 : c == 1 ? i - s1 : i - 1;
            set_error("ILU(0): zero pivot at row %lld", row); set_error_row(row); rc = KRYST_ZERO_PIVOT;
        } else { set_error("ILUP: zero diagonal in U (first at row %lld)", flag[0]); rc = KRYST_SOLVE_ERROR; }
    }
    if (rc == KRYST_OK) {
        D->GL.Ni = D->GU.Ni = Ni; D->GL.Nj = D->GU.Nj = Nj; D->GL.Nk = D->GU.Nk = Nk; D->GL.ok = D->GU.ok = true;
        rc = finish_ilu_device(pc, D);
    }
    if (verbose) fprintf(stderr, "[kryst ilu] device-side setup of a %d x %d x %d grid operator: %.1f ms\n", Ni, Nj, Nk,
                         std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    if (rc != KRYST_OK) { kryst_pc_destroy(pc); return rc; }
    *out = pc;
    return KRYST_OK;
}


// ---------------------------------------------------------------------------------------------------------------------------
// Device-side IKJ factorisation of a GENERAL sparse operator (true ILU(0) on A's pattern; the pointwise kryst-compat / Ilup(0)
// quotients need no elimination and stay parallel host loops).  ONE WAVE PER ROW: the row's entries sit in LDS, its lower entries
// are taken in stored order, and for each the pivot row's entries are spread over the lanes -- lane t looks its column up in the
// row and subtracts there (a pivot row's columns are distinct, so no two lanes touch one entry; each entry receives its updates in
// pivot order: the host loop's operations in the host loop's order, bit for bit).  Rows run concurrently as far as the dependency
// graph allows: the wave polls the "row c is final" flag of the pivot row it needs next.  Waves take the rows in DEPENDENCY-LEVEL
// order (levels of A's lower pattern, computed on the host from the row pointers and columns it has anyway: in natural order the
// few thousand rows in flight are a thin slice of the graph, 100+ levels deep, and the launch crawls): a pivot row sits at a
// smaller position, workgroups start in index order, so the lowest unfinished workgroup only waits for finished ones; a poll
// budget turns a scheduling surprise into a clean fallback to the host loop.  Rows longer than 64 entries: host loop.
__global__ __launch_bounds__(256) void ilu0_dpos_kernel(const int32_t* __restrict__ rp, const int32_t* __restrict__ col, int32_t n, int32_t* dpos, int32_t* rowdone) {
    const int32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    int32_t d = -1;
    for (int32_t k = rp[i]; k < rp[i + 1]; ++k) if (col[k] == i) d = k;
    dpos[i] = d; rowdone[i] = 0;
}
__global__ __launch_bounds__(256) void ilu0_ikj_wave_kernel(const int32_t* __restrict__ rp, const int32_t* __restrict__ col, double* w,
                                                            const int32_t* __restrict__ dpos, const int32_t* __restrict__ order, int32_t n, int32_t* rowdone,
                                                            unsigned long long* first_bad, int32_t* stalled, int budget0, int ascending) {
    __shared__ int32_t lcol_all[4 * 64];
    __shared__ double lw_all[4 * 64];
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), l = threadIdx.x & 63;
    const int32_t pos = blockIdx.x * 4 + wv;
    if (pos >= n) return;                                                  // (uniform over the wave; no workgroup barrier anywhere)
    const int32_t i = order[pos];                                          // rows in dependency-level order: a pivot row sits at a smaller position
    int32_t* const lcol = lcol_all + 64 * wv;
    double* const lw = lw_all + 64 * wv;
    const int32_t kbeg = rp[i], len = rp[i + 1] - kbeg;                    // len <= 64 (checked on the host)
    lcol[l] = l < len ? col[kbeg + l] : 0x7fffffff;
    lw[l] = l < len ? w[kbeg + l] : 0.0;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    for (int e = 0; e < len; ++e) {
        const int32_t c = lcol[e];                                         // (the same word for every lane)
        if (c >= n) continue;                                              // halo column (another rank's row): not part of this block
        if (c >= i) break;
        int budget = budget0;
        while (__hip_atomic_load(&rowdone[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
            if (--budget <= 0) break;
            __builtin_amdgcn_s_sleep(1);
        }
        if (budget <= 0) { if (l == 0) __hip_atomic_store(stalled, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }   // the host loop takes over
        // (no acquire fence: everything another wave has written -- pivot and pivot row -- is read with agent-scope loads below;
        // an agent-scope fence writes back / invalidates the whole L2 on this chip, and thirteen of them per row made the launch
        // 190 ms instead of a few)
        const int32_t kd = dpos[c];
        const double pivot = kd >= 0 ? __hip_atomic_load(&w[kd], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
        if (kd < 0 || pivot == 0.0) {
            // the sequential loop stops at the first (row, entry) that meets a zero pivot and names the PIVOT row
            if (l == 0) atomicMin(first_bad, ((unsigned long long)(unsigned)i << 32) | (unsigned)c);
            break;
        }
        const double lik = lw[e] / pivot;
        __builtin_amdgcn_wave_barrier();                                   // (every lane has read lw[e] before lane 0 replaces it)
        if (l == 0) lw[e] = lik;
        const int32_t cb = rp[c], clen = rp[c + 1] - cb;
        for (int32_t o = 0; o < clen; o += 64) {
            const bool has = o + l < clen;
            const int32_t j = has ? col[cb + o + l] : -1;
            const double wc = has ? __hip_atomic_load(&w[cb + o + l], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
            if (j > c && j < n) {
                // where column j sits in this row, if it does: the row's columns ascend (kryst_csr_create), the slots behind them hold INT_MAX --
                // six halvings of the 64 slots instead of a walk over the row (a 27-point row: 13 pivot rows x 27 compares per lane)
                // (a row block of a distributed operator stores its lower neighbour's halo columns FIRST and numbers them n + h: not ascending, walked)
                if (ascending) {
                    int p = 0;
#pragma unroll
                    for (int step = 32; step >= 1; step >>= 1) if (lcol[p + step] <= j) p += step;
                    if (lcol[p] == j) lw[p] = lw[p] - lik * wc;
                } else {
                    for (int p = 0; p < len; ++p)
                        if (lcol[p] == j) { lw[p] = lw[p] - lik * wc; break; }
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    // this row's values before its flag: write-through (agent-scope) stores, waited for, then the flag -- no L2-wide release fence
    if (l < len) __hip_atomic_store(&w[kbeg + l], lw[l], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                     // the stores have been acknowledged (a workgroup-scope release fence
    __builtin_amdgcn_wave_barrier();                                       // does not wait for vector stores on this target: the flag overtook them)
    if (l == 0) __hip_atomic_store(&rowdone[i], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (also after a failure: nobody must wait for this row)
}

// w <- the true ILU(0) factor values on A's pattern, computed on the device and copied to the host vector `w` (which holds A's
// values on entry).  *used = false when the device path did not run to completion (the caller then runs the host loop).
static int32_t ikj_on_device(kryst_csr_t a, const std::vector<int64_t>& rp, const std::vector<int32_t>& col, std::vector<double>& w, bool* used, int64_t* bad_row) {
    *used = false; *bad_row = -1;
    kryst_ctx_t ctx = a->ctx;
    const int64_t n = a->nrows, nnz = a->nnz;
    if (n == 0 || nnz == 0 || n >= (1ll << 31) - 4 || env_int("KRYST_ILU_DEVICE_SETUP", 1) == 0) return KRYST_OK;
    for (int64_t i = 0; i < n; ++i) if (rp[i + 1] - rp[i] > 64) return KRYST_OK;       // (a row must fit one wave's LDS slice)
    const bool verbose = getenv("KRYST_ILU_VERBOSE") != nullptr;
    auto tnow = [] { return std::chrono::steady_clock::now(); };
    auto t0 = tnow();
    auto lap = [&](const char* what) { if (verbose) { fprintf(stderr, "[kryst ilu]   ikj: %s %.1f ms\n", what, std::chrono::duration<double, std::milli>(tnow() - t0).count()); t0 = tnow(); } };
    // dependency levels of the lower pattern (local columns below the diagonal), rows ordered by (level, row)
    std::vector<int32_t> order((size_t)n);
    {
        std::vector<int32_t> lvl((size_t)n, 0), cnt;
        int32_t nl = 0;
        for (int64_t i = 0; i < n; ++i) {
            int32_t lv = 0;
            for (int64_t k = rp[i]; k < rp[i + 1]; ++k) { const int32_t c = col[k]; if (c < i) lv = std::max(lv, lvl[c] + 1); }   // (halo columns are >= n > i)
            lvl[i] = lv; nl = std::max(nl, lv + 1);
        }
        cnt.assign((size_t)nl + 1, 0);
        for (int64_t i = 0; i < n; ++i) cnt[lvl[i] + 1]++;
        for (int32_t q = 0; q < nl; ++q) cnt[q + 1] += cnt[q];
        for (int64_t i = 0; i < n; ++i) order[cnt[lvl[i]]++] = (int32_t)i;
    }
    lap("levels on the host");
    struct Tmp { double* w = nullptr; int32_t* dpos = nullptr; int32_t* done = nullptr; int32_t* order = nullptr; unsigned long long* bad = nullptr;
                 ~Tmp() { (void)pool_free(w); (void)pool_free(dpos); (void)pool_free(done); (void)pool_free(order); (void)pool_free(bad); } } t;
    if (pool_malloc(&t.w, sizeof(double) * (size_t)nnz) != hipSuccess || pool_malloc(&t.dpos, sizeof(int32_t) * (size_t)n) != hipSuccess ||
        pool_malloc(&t.done, sizeof(int32_t) * (size_t)n) != hipSuccess || pool_malloc(&t.order, sizeof(int32_t) * (size_t)n) != hipSuccess ||
        pool_malloc(&t.bad, 16) != hipSuccess) { (void)hipGetLastError(); return KRYST_OK; }
    KR_HIP(hipMemcpyAsync(t.order, order.data(), sizeof(int32_t) * (size_t)n, hipMemcpyHostToDevice, ctx->s_main));
    KR_HIP(hipMemcpyAsync(t.w, a->d_val, sizeof(double) * (size_t)nnz, hipMemcpyDeviceToDevice, ctx->s_main));
    KR_HIP(hipMemsetAsync(t.bad, 0xff, 8, ctx->s_main));
    KR_HIP(hipMemsetAsync(t.bad + 1, 0, 8, ctx->s_main));
    const unsigned g = (unsigned)((n + 255) / 256);
    hipLaunchKernelGGL(ilu0_dpos_kernel, dim3(g), dim3(256), 0, ctx->s_main, a->d_row_ptr, a->d_col, (int32_t)n, t.dpos, t.done);
    hipLaunchKernelGGL(ilu0_ikj_wave_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, ctx->s_main, a->d_row_ptr, a->d_col, t.w, t.dpos, t.order, (int32_t)n, t.done,
                       t.bad, reinterpret_cast<int32_t*>(t.bad + 1), std::max(1, env_int("KRYST_ILU_SETUP_POLL_BUDGET", 1 << 22)), a->dist ? 0 : 1);
    KR_HIP(hipGetLastError());
    unsigned long long flags[2] = {0, 0};
    KR_HIP(hipMemcpyAsync(flags, t.bad, 16, hipMemcpyDeviceToHost, ctx->s_main));
    KR_HIP(hipStreamSynchronize(ctx->s_main));
    lap("allocations, uploads, kernels");
    if ((int32_t)flags[1] != 0) return KRYST_OK;                       // stalled: host loop
    if (flags[0] != ~0ull) { *bad_row = (int64_t)(flags[0] & 0xffffffffull); *used = true; return KRYST_OK; }
    KR_HIP(hipMemcpyAsync(w.data(), t.w, sizeof(double) * (size_t)nnz, hipMemcpyDeviceToHost, ctx->s_main));
    KR_HIP(hipStreamSynchronize(ctx->s_main));
    lap("factor values to the host");
    *used = true;
    return KRYST_OK;
}

// ---------------------------------------------------------------------------------------------------------------------------
// GENERAL operators, everything on the device (round 3).  The host loops of kryst_pc_ilu0 download A, eliminate, split into L / U,
// level-order both factors and upload them again; here the VALUES never leave the GPU: the host gets A's pattern (row pointers and
// columns, 4 bytes per entry, once) and does the symbolic part -- dependency levels of the lower and the upper pattern, the rows by
// level, positions, row pointers -- while the device does everything that touches a value:
//   dpos / longest row -> factor values w (mode 2: ilu0_ikj_wave_kernel in the lower pattern's level order; modes 0 / 1: the pointwise
//   quotients) -> kept entries per row and the divisor (checked against the pattern: an operator whose factor DROPS entries -- stored or
//   computed zeros -- takes the host path) -> the factors written in level order straight from (A's pattern, w) -> finish_ilu_device.
// (Levels computed on the device by a sync-free launch in natural row order were tried first: half a million waiting lanes polling
// uncached flags made a 10 716-level band matrix take 1.07 s for what the host's two O(nnz) sweeps do in 40 ms.)
// Same kept entries, same stored order, same divisors as the host path: the apply is bit-identical (the tests run both).  Taken for
// operators -- a rank's diagonal block included: halo columns (>= n) are skipped everywhere -- that are not candidate grid operators
// (more than 7 distinct offsets, 9 for a distributed block; or no offset dictionary at all) and
// whose rows fit a wave (<= 64 entries); everything else, KRYST_ILU_DEVICE_SETUP=0 and any starved poll budget: the host path.
__global__ __launch_bounds__(256) void gen_rowmax_kernel(const int32_t* __restrict__ rp, int32_t n, int32_t* maxlen) {
    const int32_t i = blockIdx.x * 256 + threadIdx.x;
    int32_t len = i < n ? rp[i + 1] - rp[i] : 0;
    for (int off = 32; off >= 1; off >>= 1) len = max(len, __shfl_xor(len, off, 64));
    if ((threadIdx.x & 63) == 0 && len > 0) atomicMax(maxlen, len);
}
// modes 0 / 1 (ilu.rs:76-80, ilup.rs:104-111): l_ij = a_ij / a_jj for stored nonzeros below the diagonal, everything else as stored
__global__ __launch_bounds__(256) void gen_pointwise_kernel(const int32_t* __restrict__ rp, const int32_t* __restrict__ col, const double* __restrict__ val,
                                                            const int32_t* __restrict__ dpos, int32_t n, int mode, double* w, unsigned long long* first_bad) {
    const int32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    for (int32_t k = rp[i]; k < rp[i + 1]; ++k) {
        const int32_t j = col[k];
        if (j < i && val[k] != 0.0) {
            const double ujj = dpos[j] >= 0 ? val[dpos[j]] : 0.0;
            if (mode == KRYST_ILU_ILUP0 && ujj == 0.0) { atomicMin(first_bad, ((unsigned long long)(unsigned)i << 32) | (unsigned)j); return; }   // ilup.rs:106-108
            w[k] = val[k] / ujj;
        }
    }
}
// kept entries (`!= T::zero()` filters, local columns only) below / above the diagonal and the backward solve's divisor
__global__ __launch_bounds__(256) void gen_classify_kernel(const int32_t* __restrict__ rp, const int32_t* __restrict__ col, const double* __restrict__ w,
                                                           int32_t n, int divide, int32_t* nl, int32_t* nu, double* dg) {
    const int32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    int32_t a = 0, b = 0; double d = 1.0;
    for (int32_t k = rp[i]; k < rp[i + 1]; ++k) {
        const int32_t j = col[k];
        if (j >= n || w[k] == 0.0) continue;
        if (j < i) ++a; else if (j > i) ++b; else if (divide) d = w[k];                  // ilup.rs:160-164 (missing diagonal: no divide)
    }
    nl[i] = a; nu[i] = b; dg[i] = d;
}
// one factor in level order: position p holds row rowid[p]; its kept entries in stored order, columns as positions
// Is every off-diagonal entry of a (candidate) box operator the coupling to a neighbour inside the 3 x 3 x 3 cube of an Ni x Nj x Nk box, columns
// ascending?  Then the hyperplanes i + 2 j + 4 k are a valid elimination order (every lower entry lies on an earlier plane), and the set-up
// needs neither the pattern on the host nor its level sweeps: box_plane_hist_kernel / box_plane_order_kernel sort the rows by plane on the device.
__global__ __launch_bounds__(256) void box_check_kernel(const int32_t* __restrict__ rp, const int32_t* __restrict__ col, int32_t n, int32_t Ni, int32_t Nj, int32_t* bad) {
    const int32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int32_t ii = i % Ni, jj = (i / Ni) % Nj, kk = i / (Ni * Nj);
    int32_t prev = -1;
    for (int32_t k = rp[i]; k < rp[i + 1]; ++k) {
        const int32_t j = col[k];
        if (j >= n || j <= prev) { *bad = 1; return; }
        prev = j;
        const int di = j % Ni - ii, dj = (j / Ni) % Nj - jj, dk = j / (Ni * Nj) - kk;
        if (di < -1 || di > 1 || dj < -1 || dj > 1 || dk < -1 || dk > 1) { *bad = 1; return; }
    }
}
__global__ __launch_bounds__(256) void box_plane_hist_kernel(int32_t n, int32_t Ni, int32_t Nj, int32_t* hist) {
    const int32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) atomicAdd(&hist[i % Ni + 2 * ((i / Ni) % Nj) + 4 * (i / (Ni * Nj))], 1);
}
__global__ __launch_bounds__(256) void box_plane_order_kernel(int32_t n, int32_t Ni, int32_t Nj, int32_t* cursor, int32_t* order) {
    const int32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) order[atomicAdd(&cursor[i % Ni + 2 * ((i / Ni) % Nj) + 4 * (i / (Ni * Nj))], 1)] = i;
}

// Box-stencil factors straight from the factor values on A's pattern (round 4): stream a of row i = the entry whose column is the
// (dk, dj, di) neighbour of (i, j, k) in the Ni x Nj x Nk box -- BoxFactor's layout; `bad` is raised by an entry that is no such neighbour
// (it wraps around a line or plane end), a halo column, or a row whose columns do not ascend.  Zero values stay +0.0 = no entry.
__global__ __launch_bounds__(256) void gen_box_fill_kernel(const int32_t* __restrict__ rp, const int32_t* __restrict__ col, const double* __restrict__ w, const double* __restrict__ dg,
                                                           int32_t n, int32_t Ni, int32_t Nj, int32_t Nk, double* cl, double* cu, int64_t cs, double* diag, int32_t* bad,
                                                           unsigned long long* counts) {           // counts[0..12]: entries per L stream, [13..25]: per U stream
    __shared__ unsigned int cnt[26];
    if (threadIdx.x < 26) cnt[threadIdx.x] = 0;
    __syncthreads();
    const int32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) {
        const int32_t ii = i % Ni, jj = (i / Ni) % Nj, kk = i / (Ni * Nj);
        int32_t prev = -1;
        for (int32_t k = rp[i]; k < rp[i + 1]; ++k) {
            const int32_t j = col[k];
            if (j >= n || j <= prev) { *bad = 1; break; }
            prev = j;
            if (j == i || w[k] == 0.0) continue;
            const int di = j % Ni - ii, dj = (j / Ni) % Nj - jj, dk = j / (Ni * Nj) - kk;
            if (di < -1 || di > 1 || dj < -1 || dj > 1 || dk < -1 || dk > 1) { *bad = 1; break; }
            const int code = 9 * (dk + 1) + 3 * (dj + 1) + (di + 1);
            if (j < i) { cl[(int64_t)code * cs + i] = w[k]; atomicAdd(&cnt[code], 1u); }
            else { cu[(int64_t)(code - 14) * cs + i] = w[k]; atomicAdd(&cnt[13 + code - 14], 1u); }
        }
        diag[i] = dg[i];
    }
    __syncthreads();
    if (threadIdx.x < 26 && cnt[threadIdx.x]) atomicAdd(&counts[threadIdx.x], (unsigned long long)cnt[threadIdx.x]);
}

template <bool LOWER>
__global__ __launch_bounds__(256) void gen_fill_kernel(const int32_t* __restrict__ rp, const int32_t* __restrict__ col, const double* __restrict__ w, int32_t n,
                                                       const int32_t* __restrict__ rowid, const int32_t* __restrict__ pos, const int32_t* __restrict__ ptr,
                                                       const double* __restrict__ dg, int32_t* out_col, double* out_val, double* out_diag,
                                                       int32_t* ecol, double* eval, uint8_t* elen) {
    const int32_t p = blockIdx.x * 256 + threadIdx.x;
    if (p >= n) return;
    const int32_t i = rowid[p];
    int32_t dst = ptr[p], u = 0;
    for (int32_t k = rp[i]; k < rp[i + 1]; ++k) {
        const int32_t j = col[k];
        if (j >= n || w[k] == 0.0) continue;
        if (LOWER ? j < i : j > i) {
            out_col[dst] = pos[j]; out_val[dst] = w[k];
            if (ecol && u < ELLW) { ecol[(int64_t)u * n + p] = pos[j]; eval[(int64_t)u * n + p] = w[k]; }
            ++dst; ++u;
        }
    }
    out_diag[p] = LOWER ? 1.0 : dg[i];
    if (elen) elen[p] = (uint8_t)u;
}
__global__ __launch_bounds__(256) void gen_maplu_kernel(const int32_t* __restrict__ posL, const int32_t* __restrict__ posU, int32_t n, int32_t* mapLU) {
    const int32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) mapLU[posU[i]] = posL[i];
}

// rows ordered by (level, row); -> rowid, pos, number of levels, level offsets
static void rows_by_level(const std::vector<int32_t>& lvl, std::vector<int32_t>& rowid, std::vector<int32_t>& pos, std::vector<int32_t>& lvl_off) {
    const int64_t n = (int64_t)lvl.size();
    int32_t nlv = 0;
    for (int64_t i = 0; i < n; ++i) nlv = std::max(nlv, lvl[i] + 1);
    lvl_off.assign((size_t)nlv + 1, 0);
    for (int64_t i = 0; i < n; ++i) lvl_off[lvl[i] + 1]++;
    for (int32_t q = 0; q < nlv; ++q) lvl_off[q + 1] += lvl_off[q];
    std::vector<int32_t> cursor(lvl_off.begin(), lvl_off.end() - 1);
    rowid.resize((size_t)n); pos.resize((size_t)n);
    for (int64_t i = 0; i < n; ++i) rowid[cursor[lvl[i]]++] = (int32_t)i;        // ascending row inside a level
    for (int64_t p = 0; p < n; ++p) pos[rowid[p]] = (int32_t)p;
}

static int32_t general_setup_on_device(kryst_csr_t a, int mode, kryst_pc_t* out) {
    *out = nullptr;
    kryst_ctx_t ctx = a->ctx;
    const int64_t n64 = a->nrows, nnz = a->nnz;
    if (n64 == 0 || nnz == 0 || n64 >= (1ll << 31) - 4 || env_int("KRYST_ILU_DEVICE_SETUP", 1) == 0) return KRYST_OK;
    const int32_t n = (int32_t)n64;
    int64_t box_ni = 0, box_nj = 0;                                         // > 0: the operator's offsets are those of a box stencil on lines of box_ni rows, box_nj lines per plane
    tl_setup_stream = ctx->s_main;
    const bool verbose = getenv("KRYST_ILU_VERBOSE") != nullptr;
    auto tnow = [] { return std::chrono::steady_clock::now(); };
    auto t0 = tnow();
    auto lap = [&](const char* what) { if (verbose) { fprintf(stderr, "[kryst ilu]   device setup: %s %.1f ms\n", what, std::chrono::duration<double, std::milli>(tnow() - t0).count()); t0 = tnow(); } };
    if (a->d_code) {                         // a handful of offsets: a candidate grid operator (thin boxes, 2-D operators the device-side grid
                                             // setup passed on) -- the host path recognises those and takes the wavefront kernels
        int32_t used[256]; int32_t* d_used = nullptr;
        KR_HIP(pool_malloc(&d_used, sizeof used));
        (void)hipMemsetAsync(d_used, 0, sizeof used, ctx->s_main);
        hipLaunchKernelGGL(code_usage_kernel, dim3((unsigned)std::min<int64_t>(4096, (nnz + 255) / 256 + 1)), dim3(256), 0, ctx->s_main, a->d_code, nnz, d_used);
        const hipError_t e1 = hipMemcpyAsync(used, d_used, sizeof used, hipMemcpyDeviceToHost, ctx->s_main), e2 = hipStreamSynchronize(ctx->s_main);
        (void)pool_free(d_used);
        if (e1 != hipSuccess || e2 != hipSuccess) { (void)hipGetLastError(); return KRYST_OK; }
        int cnt = 0;
        for (int q = 0; q < 256; ++q) cnt += used[q] ? 1 : 0;
        // (a row block of a distributed grid operator lists up to two halo offsets besides its seven)
        if (cnt <= (a->dist ? 9 : 7) && env_int("KRYST_ILU_GRID", 1) != 0) return KRYST_OK;      // (with the grid forms switched off nothing would recognise it anyway)
        // a box stencil inside the 3 x 3 x 3 cube (up to 27 offsets that decompose for one Ni, Nj): the host path lays its factors out as
        // natural-order streams for the box kernels (round 4)
        if (!a->dist && cnt <= 27 && env_int("KRYST_ILU_BOX", KR_ILU_BOX_DEFAULT) != 0) {
            int32_t dict[256];
            if (hipMemcpyAsync(dict, a->d_dict, sizeof dict, hipMemcpyDeviceToHost, ctx->s_main) == hipSuccess && hipStreamSynchronize(ctx->s_main) == hipSuccess) {
                std::vector<int64_t> offs;
                for (int q = 0; q < 256; ++q) if (used[q]) offs.push_back(dict[q]);
                int64_t bi = 0, bj = 0;
                std::vector<int64_t> lower, upper;
                for (int64_t o : offs) { if (o < 0) lower.push_back(o); else if (o > 0) upper.push_back(o); }
                if (lower.size() <= 13 && upper.size() <= 13 && box_dims_from_offsets(offs, n64, &bi, &bj) && n64 >= 27) { box_ni = bi; box_nj = bj; }
            } else (void)hipGetLastError();
        }
    }
    struct Tmp {
        double *w = nullptr, *dg = nullptr; int32_t *dpos = nullptr, *done = nullptr, *order = nullptr, *nl = nullptr, *nu = nullptr;
        int32_t *posL = nullptr, *posU = nullptr; unsigned long long* flags = nullptr;
        ~Tmp() { for (void* q : {(void*)w, (void*)dg, (void*)dpos, (void*)done, (void*)order, (void*)nl, (void*)nu, (void*)posL, (void*)posU, (void*)flags}) (void)pool_free(q); }
    } t;
    const size_t nb = sizeof(int32_t) * (size_t)n;
    if (pool_malloc(&t.w, sizeof(double) * (size_t)nnz) != hipSuccess || pool_malloc(&t.dg, sizeof(double) * (size_t)n) != hipSuccess || pool_malloc(&t.dpos, nb) != hipSuccess ||
        pool_malloc(&t.done, nb) != hipSuccess || pool_malloc(&t.order, nb) != hipSuccess ||
        pool_malloc(&t.nl, nb) != hipSuccess || pool_malloc(&t.nu, nb) != hipSuccess || pool_malloc(&t.posL, nb) != hipSuccess || pool_malloc(&t.posU, nb) != hipSuccess ||
        pool_malloc(&t.flags, 64) != hipSuccess) { (void)hipGetLastError(); return KRYST_OK; }
    const unsigned g = (unsigned)((n + 255) / 256);
    const int budget = std::max(1, env_int("KRYST_ILU_SETUP_POLL_BUDGET", 1 << 22));
    // flags: [0] first zero pivot (min), [1] stalled, [2] longest row
    KR_HIP(hipMemsetAsync(t.flags, 0xff, 8, ctx->s_main));
    KR_HIP(hipMemsetAsync(t.flags + 1, 0, 56, ctx->s_main));
    int32_t* d_stalled = reinterpret_cast<int32_t*>(t.flags + 1); int32_t* d_maxlen = reinterpret_cast<int32_t*>(t.flags + 2);
    KR_HIP(hipMemcpyAsync(t.w, a->d_val, sizeof(double) * (size_t)nnz, hipMemcpyDeviceToDevice, ctx->s_main));
    hipLaunchKernelGGL(ilu0_dpos_kernel, dim3(g), dim3(256), 0, ctx->s_main, a->d_row_ptr, a->d_col, n, t.dpos, t.done);
    hipLaunchKernelGGL(gen_rowmax_kernel, dim3(g), dim3(256), 0, ctx->s_main, a->d_row_ptr, n, d_maxlen);
    unsigned long long hf[3];
    KR_HIP(hipMemcpyAsync(hf, t.flags, sizeof hf, hipMemcpyDeviceToHost, ctx->s_main));
    KR_HIP(hipStreamSynchronize(ctx->s_main));
    if ((int32_t)hf[2] > 64) return KRYST_OK;                               // a row must fit one wave's LDS slice
    // ---- a box operator whose every entry is a neighbour inside the cube: rows sorted by hyperplane on the device, no pattern on the host
    bool plane_order = false;
    if (box_ni > 0) {
        const int32_t Ni = (int32_t)box_ni, Nj = (int32_t)box_nj, Nk = (int32_t)(n64 / (box_ni * box_nj));
        const int32_t H = Ni + 2 * Nj + 4 * Nk;
        int32_t* d_hist = nullptr; int32_t bad = 1;
        int32_t* d_bad = reinterpret_cast<int32_t*>(t.flags + 4);
        std::vector<int32_t> hist((size_t)H + 1, 0);
        if (pool_malloc(&d_hist, sizeof(int32_t) * ((size_t)H + 1)) == hipSuccess && hipMemsetAsync(d_hist, 0, sizeof(int32_t) * ((size_t)H + 1), ctx->s_main) == hipSuccess) {
            hipLaunchKernelGGL(box_check_kernel, dim3(g), dim3(256), 0, ctx->s_main, a->d_row_ptr, a->d_col, n, Ni, Nj, d_bad);
            hipLaunchKernelGGL(box_plane_hist_kernel, dim3(g), dim3(256), 0, ctx->s_main, n, Ni, Nj, d_hist);
            if (hipGetLastError() == hipSuccess && hipMemcpyAsync(&bad, d_bad, sizeof bad, hipMemcpyDeviceToHost, ctx->s_main) == hipSuccess &&
                hipMemcpyAsync(hist.data(), d_hist, sizeof(int32_t) * (size_t)H, hipMemcpyDeviceToHost, ctx->s_main) == hipSuccess &&
                hipStreamSynchronize(ctx->s_main) == hipSuccess && bad == 0) {
                int32_t run = 0;
                for (int32_t h = 0; h < H; ++h) { const int32_t c = hist[(size_t)h]; hist[(size_t)h] = run; run += c; }      // exclusive scan: the planes' first positions
                if (run == n && hipMemcpyAsync(d_hist, hist.data(), sizeof(int32_t) * (size_t)H, hipMemcpyHostToDevice, ctx->s_main) == hipSuccess) {
                    hipLaunchKernelGGL(box_plane_order_kernel, dim3(g), dim3(256), 0, ctx->s_main, n, Ni, Nj, d_hist, t.order);
                    plane_order = hipGetLastError() == hipSuccess && hipStreamSynchronize(ctx->s_main) == hipSuccess;
                }
            }
        }
        (void)hipGetLastError();
        (void)pool_free(d_hist);
        if (!plane_order) box_ni = 0;                                         // not (provably) a box operator: the general path below
        else lap("box operator checked, rows sorted by hyperplane");
    }
    // ---- the pattern on the host: levels of the lower / upper pattern, local entries per row
    hvec<int32_t> hrp, hcol;                                               // (copied into: not value-initialised)
    std::vector<int32_t> lvl, lvlU, cntL, cntU, rowid, pos, lvl_off;
    if (!plane_order) {
    janitor_wait();
    hrp.resize((size_t)n + 1); hcol.resize((size_t)nnz); lvl.resize((size_t)n); lvlU.resize((size_t)n); cntL.resize((size_t)n); cntU.resize((size_t)n);
    KR_HIP(hipMemcpyAsync(hrp.data(), a->d_row_ptr, sizeof(int32_t) * ((size_t)n + 1), hipMemcpyDeviceToHost, ctx->s_main));
    KR_HIP(hipMemcpyAsync(hcol.data(), a->d_col, sizeof(int32_t) * (size_t)nnz, hipMemcpyDeviceToHost, ctx->s_main));
    KR_HIP(hipStreamSynchronize(ctx->s_main));
    {   // the two sweeps do not touch each other's arrays: side by side
        std::thread upper([&] {
            for (int32_t i = n - 1; i >= 0; --i) {
                int32_t lv = 0, c = 0;
                for (int32_t k = hrp[i]; k < hrp[i + 1]; ++k) { const int32_t j = hcol[k]; if (j > i && j < n) { lv = std::max(lv, lvlU[j] + 1); ++c; } }
                lvlU[i] = lv; cntU[i] = c;
            }
        });
        for (int32_t i = 0; i < n; ++i) {
            int32_t lv = 0, c = 0;
            for (int32_t k = hrp[i]; k < hrp[i + 1]; ++k) { const int32_t j = hcol[k]; if (j < i) { lv = std::max(lv, lvl[j] + 1); ++c; } }
            lvl[i] = lv; cntL[i] = c;
        }
        upper.join();
    }
    lap("pattern to the host, levels");
    }
    // ---- factor values
    if (mode == KRYST_ILU_TRUE_ILU0) {
        if (!plane_order) {
            rows_by_level(lvl, rowid, pos, lvl_off);
            KR_HIP(hipMemcpyAsync(t.order, rowid.data(), nb, hipMemcpyHostToDevice, ctx->s_main));
        }
        hipLaunchKernelGGL(ilu0_ikj_wave_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, ctx->s_main, a->d_row_ptr, a->d_col, t.w, t.dpos, t.order, n, t.done,
                           t.flags, d_stalled, budget, a->dist ? 0 : 1);
    } else {
        hipLaunchKernelGGL(gen_pointwise_kernel, dim3(g), dim3(256), 0, ctx->s_main, a->d_row_ptr, a->d_col, a->d_val, t.dpos, n, mode, t.w, t.flags);
    }
    KR_HIP(hipGetLastError());
    // ---- kept entries and divisors; a factor that drops entries does not have the pattern's levels: host path
    const int divide = mode != KRYST_ILU_KRYST_COMPAT ? 1 : 0;              // ilu.rs:115-119 never divides
    hipLaunchKernelGGL(gen_classify_kernel, dim3(g), dim3(256), 0, ctx->s_main, a->d_row_ptr, a->d_col, t.w, n, divide, t.nl, t.nu, t.dg);
    KR_HIP(hipGetLastError());
    std::vector<int32_t> nl((size_t)n), nu((size_t)n);
    KR_HIP(hipMemcpyAsync(nl.data(), t.nl, nb, hipMemcpyDeviceToHost, ctx->s_main));
    KR_HIP(hipMemcpyAsync(nu.data(), t.nu, nb, hipMemcpyDeviceToHost, ctx->s_main));
    KR_HIP(hipMemcpyAsync(hf, t.flags, sizeof hf, hipMemcpyDeviceToHost, ctx->s_main));
    KR_HIP(hipStreamSynchronize(ctx->s_main));
    if ((int32_t)hf[1] != 0) return KRYST_OK;                               // a starved poll budget: the host path
    if (hf[0] != ~0ull) {
        if (mode == KRYST_ILU_TRUE_ILU0) { const long long c = (long long)(hf[0] & 0xffffffffull); set_error("ILU(0): zero pivot at row %lld", c); set_error_row(c); return KRYST_ZERO_PIVOT; }
        set_error("ILUP: zero diagonal in U at row %lld", (long long)(hf[0] & 0xffffffffull));
        return KRYST_SOLVE_ERROR;
    }
    lap("factor values, kept entries");
    kryst_pc_t pc = new kryst_pc_s();
    pc->ctx = ctx; pc->kind = KR_PC_ILU; pc->a = a; pc->n = n; pc->ilu_mode = mode; pc->divide_diag = divide;
    IluData* D = new IluData();
    D->n = n;
    pc->d_work = reinterpret_cast<double*>(D);
    int32_t rc = KRYST_OK;
    if (box_ni > 0) {
        // ---- box stencil: 13 natural-order coefficient streams per factor, written by one kernel (no level machinery at all)
        const size_t cb = sizeof(double) * (size_t)13 * (size_t)box_stream_stride(n);
        int32_t* d_bad = reinterpret_cast<int32_t*>(t.flags + 3);
        unsigned long long* d_counts = nullptr; unsigned long long counts[26];
        int32_t bad = 1;
        if (pool_malloc(&D->BL.d_c, cb) != hipSuccess || pool_malloc(&D->BU.d_c, cb) != hipSuccess || pool_malloc(&D->BU.d_diag, sizeof(double) * (size_t)n) != hipSuccess) {
            (void)hipGetLastError();
        } else if (pool_malloc(&d_counts, sizeof counts) == hipSuccess && hipMemsetAsync(d_counts, 0, sizeof counts, ctx->s_main) == hipSuccess &&
                   hipMemsetAsync(D->BL.d_c, 0, cb, ctx->s_main) == hipSuccess && hipMemsetAsync(D->BU.d_c, 0, cb, ctx->s_main) == hipSuccess) {
            const int32_t Ni = (int32_t)box_ni, Nj = (int32_t)box_nj, Nk = (int32_t)(n64 / (box_ni * box_nj));
            hipLaunchKernelGGL(gen_box_fill_kernel, dim3(g), dim3(256), 0, ctx->s_main, a->d_row_ptr, a->d_col, t.w, t.dg, n, Ni, Nj, Nk, D->BL.d_c, D->BU.d_c, box_stream_stride(n), D->BU.d_diag, d_bad,
                               d_counts);
            if (hipGetLastError() == hipSuccess && hipMemcpyAsync(&bad, d_bad, sizeof bad, hipMemcpyDeviceToHost, ctx->s_main) == hipSuccess &&
                hipMemcpyAsync(counts, d_counts, sizeof counts, hipMemcpyDeviceToHost, ctx->s_main) == hipSuccess &&
                hipStreamSynchronize(ctx->s_main) == hipSuccess && bad == 0) {
                for (BoxFactor* F : {&D->BL, &D->BU}) { F->Ni = Ni; F->Nj = Nj; F->Nk = Nk; F->ok = true; }
                box_classify(counts, true, Ni, Nj, Nk, &D->BL); box_classify(counts + 13, false, Ni, Nj, Nk, &D->BU);
            }
        }
        (void)pool_free(d_counts);
        (void)hipGetLastError();
        if (D->BL.ok && D->BU.ok) {
            lap("box-stencil streams");
            rc = finish_ilu_device(pc, D);
            if (rc != KRYST_OK) { kryst_pc_destroy(pc); return rc; }
            *out = pc;
            return KRYST_OK;
        }
        D->BL.free_all(); D->BU.free_all(); D->BL = BoxFactor(); D->BU = BoxFactor();       // not a box operator after all: the level-ordered forms
        if (plane_order) { kryst_pc_destroy(pc); return KRYST_OK; }                           // (no pattern on the host to level-order from: the host path)
    }
    if (nl != cntL || nu != cntU) { kryst_pc_destroy(pc); return KRYST_OK; }
    // ---- the two level-ordered factors
    std::vector<int32_t> ptr((size_t)n + 1);
    for (int which = 0; which < 2 && rc == KRYST_OK; ++which) {
        TriFactor* F = which == 0 ? &D->L : &D->U;
        const std::vector<int32_t>& lv = which == 0 ? lvl : lvlU;
        const std::vector<int32_t>& len = which == 0 ? nl : nu;
        rows_by_level(lv, rowid, pos, F->lvl_off);
        int32_t maxlen = 0;
        ptr[0] = 0;
        int64_t nlong = 0;
        for (int32_t p = 0; p < n; ++p) { const int32_t L = len[rowid[p]]; ptr[(size_t)p + 1] = ptr[p] + L; maxlen = std::max(maxlen, L); nlong += L > 8; }
        const size_t fn = (size_t)ptr[n];
        F->npos = n;
        F->ell = maxlen <= ELLW;
        F->held = maxlen <= 8 ? 8 : 16;
        const double rows_per_level = (double)n / (double)std::max<size_t>(1, F->lvl_off.size() - 1);
        F->syncfree = env_int("KRYST_ILU_SYNCFREE", ((F->ell && (rows_per_level >= 256.0 || env_int("KRYST_ILU_RUN_FREE", 1) == 0)) || rows_per_level >= 512.0) ? 1 : 0) != 0;
        F->last_entry = (int32_t)fn - 1;
        F->free_runs = !F->syncfree && fn > 0 && grant_free_runs(n > 0 ? (double)nlong / (double)n : 0.0);
        int32_t* d_pos = which == 0 ? t.posL : t.posU;
        rc = up(&F->d_row, rowid);
        if (rc == KRYST_OK) rc = up(&F->d_ptr, ptr);
        if (rc == KRYST_OK) rc = up(&F->d_lvl_off, F->lvl_off);
        if (rc == KRYST_OK && (pool_malloc(&F->d_col, sizeof(int32_t) * (fn + 1)) != hipSuccess || pool_malloc(&F->d_val, sizeof(double) * (fn + 1)) != hipSuccess ||
                               pool_malloc(&F->d_diag, sizeof(double) * ((size_t)n + 1)) != hipSuccess)) { set_error("hipMalloc failed"); rc = KRYST_ERR_HIP; }
        if (rc == KRYST_OK && F->ell) {
            const size_t eb = (size_t)ELLW * n;
            if (pool_malloc(&F->d_ecol, sizeof(int32_t) * (eb + 1)) != hipSuccess || pool_malloc(&F->d_eval, sizeof(double) * (eb + 1)) != hipSuccess ||
                pool_malloc(&F->d_elen, (size_t)n + 1) != hipSuccess) { set_error("hipMalloc failed"); rc = KRYST_ERR_HIP; }
            else if (hipMemsetAsync(F->d_ecol, 0, sizeof(int32_t) * (eb + 1), ctx->s_main) != hipSuccess ||
                     hipMemsetAsync(F->d_eval, 0, sizeof(double) * (eb + 1), ctx->s_main) != hipSuccess) rc = KRYST_ERR_HIP;      // padding slots: column 0, value 0
        }
        if (rc == KRYST_OK && hipMemcpyAsync(d_pos, pos.data(), nb, hipMemcpyHostToDevice, ctx->s_main) != hipSuccess) rc = KRYST_ERR_HIP;
        if (rc == KRYST_OK) {
            if (which == 0) hipLaunchKernelGGL((gen_fill_kernel<true>), dim3(g), dim3(256), 0, ctx->s_main, a->d_row_ptr, a->d_col, t.w, n, F->d_row, d_pos, F->d_ptr, t.dg,
                                               F->d_col, F->d_val, F->d_diag, F->ell ? F->d_ecol : nullptr, F->d_eval, F->ell ? F->d_elen : nullptr);
            else hipLaunchKernelGGL((gen_fill_kernel<false>), dim3(g), dim3(256), 0, ctx->s_main, a->d_row_ptr, a->d_col, t.w, n, F->d_row, d_pos, F->d_ptr, t.dg,
                                    F->d_col, F->d_val, F->d_diag, F->ell ? F->d_ecol : nullptr, F->d_eval, F->ell ? F->d_elen : nullptr);
            if (hipGetLastError() != hipSuccess || hipStreamSynchronize(ctx->s_main) != hipSuccess) { set_error("factor fill kernel failed"); rc = KRYST_ERR_HIP; }   // (pos lives in a host vector reused below)
            if (rc == KRYST_OK) rc = build_free_streams(F, ctx->s_main, ptr.data());
        }
    }
    if (rc == KRYST_OK && pool_malloc(&D->d_mapLU, nb + 4) != hipSuccess) { set_error("hipMalloc failed"); rc = KRYST_ERR_HIP; }
    if (rc == KRYST_OK) {
        hipLaunchKernelGGL(gen_maplu_kernel, dim3(g), dim3(256), 0, ctx->s_main, t.posL, t.posU, n, D->d_mapLU);
        if (hipGetLastError() != hipSuccess || hipStreamSynchronize(ctx->s_main) != hipSuccess) rc = KRYST_ERR_HIP;
    }
    lap("level-ordered factors");
    if (rc == KRYST_OK) rc = finish_ilu_device(pc, D);
    if (rc != KRYST_OK) { kryst_pc_destroy(pc); return rc; }
    *out = pc;
    return KRYST_OK;
}

static int32_t download_rows(kryst_csr_t a, hvec<int64_t>& rp, hvec<int32_t>& col, hvec<double>& val) {
    kr::janitor_wait();
    rp.resize((size_t)a->nrows + 1); col.resize((size_t)a->nnz); val.resize((size_t)a->nnz);
    return kryst_csr_download(a, rp.data(), col.data(), val.data());
}


extern "C" int32_t kryst_pc_ilu0(kryst_csr_t a, int32_t mode, kryst_pc_t* out) {
    KR_ARG(a && out && mode >= 0 && mode <= 2, "pc_ilu0");
    KR_ARG(a->nrows == a->xlen, "pc_ilu0: square operator required");
    kryst_ctx_t ctx = a->ctx;
    KR_HIP(hipSetDevice(ctx->device));
    {   // grid operators are factored on the device (no download, no host loops)
        kryst_pc_t pc = nullptr;
        KR_TRY(grid_setup_on_device(a, mode, &pc));
        if (pc) { *out = pc; return KRYST_OK; }
        KR_TRY(general_setup_on_device(a, mode, &pc));      // general operators: factor values, levels and the level-ordered factors on the device
        if (pc) { *out = pc; return KRYST_OK; }
    }
    const int64_t n = a->nrows, nnz = a->nnz;
    const bool verbose = getenv("KRYST_ILU_VERBOSE") != nullptr;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto ms_since = [&](std::chrono::steady_clock::time_point t0) { return std::chrono::duration<double, std::milli>(now() - t0).count(); };
    auto t_phase = now();
    std::vector<int64_t> rp((size_t)n + 1); std::vector<int32_t> col((size_t)nnz); std::vector<double> val((size_t)nnz);
    KR_TRY(kryst_csr_download(a, rp.data(), col.data(), val.data()));
    if (verbose) { fprintf(stderr, "[kryst ilu] download %.0f ms\n", ms_since(t_phase)); t_phase = now(); }
    // diagonal position of every row (halo columns, col >= n, are outside the local block)
    std::vector<int64_t> dpos((size_t)n, -1);
    par_rows(n, [&](int64_t lo, int64_t hi) {
        for (int64_t i = lo; i < hi; ++i)
            for (int64_t k = rp[i]; k < rp[i + 1]; ++k) if (col[k] == i) dpos[i] = k;
    });
    std::vector<double> w(val);                       // factor values on A's pattern
    bool on_device = false;
    if (mode == KRYST_ILU_TRUE_ILU0) {                // IKJ restricted to the pattern: on the device (one lane per row, rows concurrent
        int64_t bad = -1;                             // along the dependency graph), same operations in the same order as the loop below
        KR_TRY(ikj_on_device(a, rp, col, w, &on_device, &bad));
        if (on_device && bad >= 0) { set_error("ILU(0): zero pivot at row %lld", (long long)bad); set_error_row(bad); return KRYST_ZERO_PIVOT; }
        if (verbose && on_device) { fprintf(stderr, "[kryst ilu] device-side IKJ factorisation %.0f ms\n", ms_since(t_phase)); t_phase = now(); }
    }
    if (mode == KRYST_ILU_TRUE_ILU0 && !on_device) {  // host loop (KRYST_ILU_DEVICE_SETUP=0, or the device path gave up)
        std::vector<int64_t> pos((size_t)n, -1);
        for (int64_t i = 0; i < n; ++i) {
            for (int64_t k = rp[i]; k < rp[i + 1]; ++k) if (col[k] < n) pos[col[k]] = k;
            for (int64_t k = rp[i]; k < rp[i + 1]; ++k) {
                const int64_t c = col[k];
                if (c >= n) continue;                 // halo column (another rank's row): not part of this rank's block.  A LOWER
                if (c >= i) break;                    // neighbour's halo column is stored first and numbered n + h: skip, do not stop
                const int64_t kd = dpos[c];
                if (kd < 0 || w[kd] == 0.0) { set_error("ILU(0): zero pivot at row %lld", (long long)c); set_error_row(c); return KRYST_ZERO_PIVOT; }
                w[k] = w[k] / w[kd];
                for (int64_t kk = rp[c]; kk < rp[c + 1]; ++kk) {
                    const int64_t j = col[kk];
                    if (j > c && j < n && pos[j] >= 0) w[pos[j]] = w[pos[j]] - w[k] * w[kk];
                }
            }
            for (int64_t k = rp[i]; k < rp[i + 1]; ++k) if (col[k] < n) pos[col[k]] = -1;
        }
    } else if (mode != KRYST_ILU_TRUE_ILU0) {
        // ilu.rs:76-80 / ilup.rs:104-111: l_ij = a_ij / a_jj for stored nonzeros below the diagonal; U = triu(A)
        std::atomic<long long> bad{-1};                                    // lowest row whose pivot is zero (the reference stops at the first)
        par_rows(n, [&](int64_t lo, int64_t hi) {
            for (int64_t i = lo; i < hi; ++i)
                for (int64_t k = rp[i]; k < rp[i + 1]; ++k) {
                    const int64_t j = col[k];
                    if (j < i && val[k] != 0.0) {
                        const double ujj = dpos[j] >= 0 ? val[dpos[j]] : 0.0;
                        if (mode == KRYST_ILU_ILUP0 && ujj == 0.0) {       // ilup.rs:106-108
                            long long cur = bad.load();
                            while ((cur < 0 || (long long)i < cur) && !bad.compare_exchange_weak(cur, (long long)i)) {}
                            return;
                        }
                        w[k] = val[k] / ujj;
                    }
                }
        });
        if (bad.load() >= 0) {
            const int64_t i = bad.load();
            for (int64_t k = rp[i]; k < rp[i + 1]; ++k) {
                const int64_t j = col[k];
                if (j < i && val[k] != 0.0 && (dpos[j] < 0 || val[dpos[j]] == 0.0)) { set_error("ILUP: zero diagonal in U at row %lld", (long long)j); break; }
            }
            return KRYST_SOLVE_ERROR;
        }
    }
    if (verbose) { fprintf(stderr, "[kryst ilu] factorisation %.0f ms\n", ms_since(t_phase)); t_phase = now(); }
    const bool divide = mode != KRYST_ILU_KRYST_COMPAT;                    // ilu.rs:115-119 never divides
    FlatRows le, ue;
    le.ptr.assign((size_t)n + 1, 0); ue.ptr.assign((size_t)n + 1, 0);
    hvec<double> dg((size_t)n, 1.0);
    par_rows(n, [&](int64_t lo, int64_t hi) {                              // count, prefix, then fill (stored order = ascending column)
        for (int64_t i = lo; i < hi; ++i) {
            int64_t nl = 0, nu = 0;
            for (int64_t k = rp[i]; k < rp[i + 1]; ++k) {
                const int64_t j = col[k];
                if (j >= n || w[k] == 0.0) continue;                       // halo column / `!= T::zero()` filters
                if (j < i) ++nl; else if (j > i) ++nu;
            }
            le.ptr[i + 1] = nl; ue.ptr[i + 1] = nu;
        }
    });
    for (int64_t i = 0; i < n; ++i) { le.ptr[i + 1] += le.ptr[i]; ue.ptr[i + 1] += ue.ptr[i]; }
    le.col.resize((size_t)le.ptr[n]); le.val.resize((size_t)le.ptr[n]); ue.col.resize((size_t)ue.ptr[n]); ue.val.resize((size_t)ue.ptr[n]);
    par_rows(n, [&](int64_t lo, int64_t hi) {
        for (int64_t i = lo; i < hi; ++i) {
            int64_t wl = le.ptr[i], wu = ue.ptr[i];
            for (int64_t k = rp[i]; k < rp[i + 1]; ++k) {
                const int64_t j = col[k];
                if (j >= n || w[k] == 0.0) continue;
                if (j < i) { le.col[wl] = (int32_t)j; le.val[wl] = w[k]; ++wl; }
                else if (j > i) { ue.col[wu] = (int32_t)j; ue.val[wu] = w[k]; ++wu; }
                else if (divide) dg[i] = w[k];                             // ilup.rs:160-164 (missing diagonal: no divide)
            }
        }
    });
    if (verbose) { fprintf(stderr, "[kryst ilu] split into L / U rows %.0f ms\n", ms_since(t_phase)); t_phase = now(); }
    const int32_t rc = finish_ilu_pc(a, mode, divide, le, ue, dg, out);
    if (verbose) fprintf(stderr, "[kryst ilu] device structures %.0f ms\n", ms_since(t_phase));
    return rc;
}

// Ilup::new(fill).setup(a) exactly as written (src/preconditioner/ilup.rs:77-134), on sparse rows instead of the reference's
// dense n x n `level` / `a_work` arrays: an entry that the dense code never touches is (0.0, usize::MAX) here too.
// Ilup(p >= 1), ilup.rs:77-167: IKJ elimination with level-of-fill bookkeeping; whether an entry takes part depends on its VALUE too
// (`!= 0.0` tests at :106, :117, :129), so pattern and values are computed together, row by row.  Row i only needs the finished rows j < i
// that appear in its working row (original entries and fill): rows are dealt out to the host's cores in small blocks, round-robin,
// every thread walks its blocks in ascending order and waits on the block's "finished" flag before it uses another thread's pivot row (round 4;
// one thread did all of it before: 300 ms of the 620 ms Ilup(1) setup at 128^3).  The lowest unfinished row never waits for an
// unfinished one, so somebody always makes progress.  The same operations on the same operands in the same order as the one-thread
// loop: same bits.
static int32_t ilup_setup(kryst_csr_t a, int32_t fill, kryst_pc_t* out);
extern "C" int32_t kryst_pc_ilup(kryst_csr_t a, int32_t fill, kryst_pc_t* out) {
    const auto t0 = std::chrono::steady_clock::now();
    const int32_t rc = ilup_setup(a, fill, out);
    if (getenv("KRYST_ILU_VERBOSE")) fprintf(stderr, "[kryst ilup] all of it, host arrays freed: %.0f ms\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    return rc;
}
static int32_t ilup_setup(kryst_csr_t a, int32_t fill, kryst_pc_t* out) {
    KR_ARG(a && out && fill >= 0, "pc_ilup");
    KR_ARG(a->nrows == a->xlen, "pc_ilup: square operator required");
    if (fill == 0) return kryst_pc_ilu0(a, KRYST_ILU_ILUP0, out);
    KR_HIP(hipSetDevice(a->ctx->device));
    const int64_t n = a->nrows;
    const bool verbose = getenv("KRYST_ILU_VERBOSE") != nullptr;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto t_phase = now();
    auto lap = [&](const char* what) { if (verbose) { fprintf(stderr, "[kryst ilup] %s %.0f ms\n", what, std::chrono::duration<double, std::milli>(now() - t_phase).count()); t_phase = now(); } };
    hvec<int64_t> rp; hvec<int32_t> col; hvec<double> val;
    KR_TRY(download_rows(a, rp, col, val));
    lap("download");
    // the elimination itself: the row pipeline over the host's cores (host_factor.cpp: host_ilup_rows -- ilup.rs:77-134 on sparse rows)
    IlupOptions opt;
    opt.threads = env_int("KRYST_ILUP_THREADS", 0); opt.block = std::max(1, env_int("KRYST_ILUP_BLOCK", 2048));
    opt.cpu_group = env_int("KRYST_ILUP_CPU_GROUP", 16); opt.verbose = verbose; opt.trace = verbose && getenv("KRYST_ILUP_TRACE") != nullptr;
    FlatRows le, ue; hvec<double> dg;
    long long zero_col = -1;
    std::shared_ptr<void> scratch;
    int hrc = 2;
    try { hrc = host_ilup_rows(n, rp.data(), col.data(), val.data(), fill, opt, le, ue, dg, &zero_col, &scratch); }
    catch (const std::bad_alloc&) { hrc = 2; }
    if (hrc == 1) { set_error("ILUP: zero diagonal in U at row %lld", zero_col); return KRYST_SOLVE_ERROR; }
    if (hrc != 0) { set_error("ILUP: out of host memory during the elimination"); return KRYST_ERR_ARG; }
    const int32_t rc = finish_ilu_pc(a, 10 + fill, true, le, ue, dg, out);
    lap("device structures");
    // ~1 GB of host arrays at 128^3: unmapping them takes 60-70 ms, which the caller need not wait for
    struct Bundle { std::shared_ptr<void> scratch; FlatRows le, ue; hvec<int64_t> rp; hvec<int32_t> col; hvec<double> val; hvec<double> dg; };
    auto bundle = std::make_shared<Bundle>();
    bundle->scratch.swap(scratch);
    bundle->le.ptr.swap(le.ptr); bundle->le.col.swap(le.col); bundle->le.val.swap(le.val); bundle->ue.ptr.swap(ue.ptr); bundle->ue.col.swap(ue.col); bundle->ue.val.swap(ue.val);
    bundle->rp.swap(rp); bundle->col.swap(col); bundle->val.swap(val); bundle->dg.swap(dg);
    janitor_run(bundle);
    lap("host arrays handed to the janitor thread");
    return rc;
}

extern "C" int32_t kryst_pc_ilut(kryst_csr_t a, int32_t fill, double droptol, kryst_pc_t* out) {
    KR_ARG(a && out && fill >= 0, "pc_ilut");
    KR_ARG(a->nrows == a->xlen, "pc_ilut: square operator required");
    KR_HIP(hipSetDevice(a->ctx->device));
    const int64_t n = a->nrows;
    hvec<int64_t> rp; hvec<int32_t> col; hvec<double> val;
    KR_TRY(download_rows(a, rp, col, val));
    FlatRows le, ue; hvec<double> dg;
    try { host_ilut_rows(n, rp.data(), col.data(), val.data(), fill, droptol, le, ue, dg); }     // (host_factor.cpp; ilut.rs:80-150)
    catch (const std::bad_alloc&) { set_error("ILUT: out of host memory"); return KRYST_ERR_ARG; }
    return finish_ilu_pc(a, 100, true, le, ue, dg, out);
}

// ---- the host-side factorisations on plain host arrays (kryst_hip.h: no device, no context): what kryst_pc_ilup / kryst_pc_ilut run between the
// download of the operator's rows and the upload of the factors, for CPU-only callers, for tests against the oracle and for the sanitizer tier
struct kryst_host_factors_s { kr::FlatRows le, ue; kr::hvec<double> dg; int64_t n = 0; };
static int32_t host_rows_check(int64_t n, const int64_t* row_ptr, const int32_t* col, const double* val, const char* who) {
    if (n < 0 || !row_ptr || (row_ptr[n] > 0 && (!col || !val)) || row_ptr[0] != 0) { set_error("bad argument: %s", who); return KRYST_ERR_ARG; }
    for (int64_t i = 0; i < n; ++i) if (row_ptr[i + 1] < row_ptr[i]) { set_error("bad argument: %s: row_ptr must not decrease", who); return KRYST_ERR_ARG; }
    for (int64_t k = 0; k < row_ptr[n]; ++k) if (col[k] < 0) { set_error("bad argument: %s: negative column", who); return KRYST_ERR_ARG; }
    return KRYST_OK;
}
extern "C" int32_t kryst_host_ilup(int64_t n, const int64_t* row_ptr, const int32_t* col, const double* val, int32_t fill, int32_t threads, int64_t block,
                                   kryst_host_factors_t* out) {
    KR_ARG(out && fill >= 0, "host_ilup");
    KR_TRY(host_rows_check(n, row_ptr, col, val, "host_ilup"));
    std::unique_ptr<kryst_host_factors_s> F(new kryst_host_factors_s());
    F->n = n;
    IlupOptions opt; opt.threads = threads; opt.block = block > 0 ? block : 2048; opt.cpu_group = 0;
    long long zero_col = -1;
    int hrc = 2;
    try { hrc = host_ilup_rows(n, row_ptr, col, val, fill, opt, F->le, F->ue, F->dg, &zero_col, nullptr); } catch (const std::bad_alloc&) { hrc = 2; }
    if (hrc == 1) { set_error("ILUP: zero diagonal in U at row %lld", zero_col); set_error_row(zero_col); return KRYST_SOLVE_ERROR; }
    if (hrc != 0) { set_error("host_ilup: out of host memory"); return KRYST_ERR_ARG; }
    *out = F.release();
    return KRYST_OK;
}
extern "C" int32_t kryst_host_ilut(int64_t n, const int64_t* row_ptr, const int32_t* col, const double* val, int32_t fill, double droptol, int32_t threads,
                                   kryst_host_factors_t* out) {
    KR_ARG(out && fill >= 0, "host_ilut");
    KR_TRY(host_rows_check(n, row_ptr, col, val, "host_ilut"));
    std::unique_ptr<kryst_host_factors_s> F(new kryst_host_factors_s());
    F->n = n;
    try { host_ilut_rows(n, row_ptr, col, val, fill, droptol, F->le, F->ue, F->dg, threads); } catch (const std::bad_alloc&) { set_error("host_ilut: out of host memory"); return KRYST_ERR_ARG; }
    *out = F.release();
    return KRYST_OK;
}
extern "C" int32_t kryst_host_factors_sizes(kryst_host_factors_t f, int64_t* n, int64_t* nnz_l, int64_t* nnz_u) {
    KR_ARG(f, "host_factors_sizes");
    if (n) *n = f->n;
    if (nnz_l) *nnz_l = f->le.ptr.empty() ? 0 : f->le.ptr[(size_t)f->n];
    if (nnz_u) *nnz_u = f->ue.ptr.empty() ? 0 : f->ue.ptr[(size_t)f->n];
    return KRYST_OK;
}
extern "C" int32_t kryst_host_factors_get(kryst_host_factors_t f, int64_t* l_ptr, int32_t* l_col, double* l_val, int64_t* u_ptr, int32_t* u_col, double* u_val, double* diag) {
    KR_ARG(f, "host_factors_get");
    const size_t n = (size_t)f->n;
    if (l_ptr) std::copy(f->le.ptr.begin(), f->le.ptr.begin() + (std::ptrdiff_t)n + 1, l_ptr);
    if (u_ptr) std::copy(f->ue.ptr.begin(), f->ue.ptr.begin() + (std::ptrdiff_t)n + 1, u_ptr);
    if (l_col) std::copy(f->le.col.begin(), f->le.col.end(), l_col);
    if (l_val) std::copy(f->le.val.begin(), f->le.val.end(), l_val);
    if (u_col) std::copy(f->ue.col.begin(), f->ue.col.end(), u_col);
    if (u_val) std::copy(f->ue.val.begin(), f->ue.val.end(), u_val);
    if (diag) std::copy(f->dg.begin(), f->dg.begin() + (std::ptrdiff_t)n, diag);
    return KRYST_OK;
}
extern "C" int32_t kryst_host_factors_destroy(kryst_host_factors_t f) { delete f; return KRYST_OK; }
extern "C" int32_t kryst_host_levels(int64_t n, const int64_t* ptr, const int32_t* col, int32_t forward, int32_t* level, int32_t* nlevels) {
    KR_ARG(n >= 0 && ptr && level && (ptr[n] == 0 || col), "host_levels");
    for (int64_t i = 0; i < n; ++i)
        for (int64_t k = ptr[i]; k < ptr[i + 1]; ++k)
            if (col[k] < 0 || col[k] >= n || (forward ? col[k] >= i : col[k] <= i)) { set_error("bad argument: host_levels: row %lld is not strictly %s", (long long)i, forward ? "lower" : "upper"); return KRYST_ERR_ARG; }
    const int32_t nl = host_levels(n, ptr, col, forward != 0, level);
    if (nlevels) *nlevels = nl;
    return KRYST_OK;
}

// What an ILU-family preconditioner's apply runs and streams (bench.py prices the triangular solve with it):
//   info[0] form: 0 level-ordered factors (sync-free / level kernels), 1 structured grid, 8 x 8 lines per workgroup, 2 structured
//           grid, 16 x 16 lines per workgroup (tri_quad.h), 3 structured grid after a give-up (plane kernels), 4 box stencil inside the
//           3 x 3 x 3 cube, one launch per hyperplane i + 2 j + 4 k (info[4], info[5]: hyperplanes per factor)
//   info[1..3] Ni, Nj, Nk (grid forms)          info[4], info[5] dependency levels of L, U (level-ordered forms)
//   box forms (4, 5): info[6], info[7] coefficient streams L / U have (of 13), info[8] 1 when both factors are "regular" (tri_box.h)
//   info[6], info[7] coefficient chunks (per block quadrant) of the forward / backward factor; info[8], info[9] how many of them
//           repeat chunk - 3 bit for bit and are not requested; info[10], info[11] bytes per chunk request (forward, backward)
//   info[12] stored entries of L + U (level-ordered forms)
extern "C" int32_t kryst_pc_ilu_info(kryst_pc_t pc, int64_t* info, int32_t count) {
    KR_ARG(pc && info && count >= 13, "pc_ilu_info: need room for 13 values");
    KR_ARG(pc->kind == KR_PC_ILU && pc->d_work, "pc_ilu_info: not an ILU-family preconditioner");
    IluData* D = reinterpret_cast<IluData*>(pc->d_work);
    kryst_ctx_t ctx = pc->ctx;
    for (int i = 0; i < count; ++i) info[i] = 0;
    if (D->BL.ok && D->BU.ok) {                                            // box stencil (round 4): 4 = one launch per hyperplane i + 2 j + 4 k
        info[0] = box_takes_wavefront(D) ? 5 : 4;                          // 5 = pipelined wavefront over parallelograms of 8 x 8 lines (tri_box.h)
        info[1] = D->BL.Ni; info[2] = D->BL.Nj; info[3] = D->BL.Nk;
        info[4] = info[5] = (D->BL.Ni - 1) + 2 * (D->BL.Nj - 1) + 4 * (D->BL.Nk - 1) + 1;
        info[6] = __builtin_popcount(D->BL.present); info[7] = __builtin_popcount(D->BU.present);   // coefficient streams the factors have (of 13 each)
        info[8] = (D->BL.regular && D->BU.regular) ? 1 : 0;                                          // tri_box.h: REGULAR
        return KRYST_OK;
    }
    if (!(D->GL.ok && D->GU.ok)) {
        info[0] = 0;
        info[4] = D->L.lvl_off.empty() ? 0 : (int64_t)D->L.lvl_off.size() - 1;
        info[5] = D->U.lvl_off.empty() ? 0 : (int64_t)D->U.lvl_off.size() - 1;
        return KRYST_OK;
    }
    const unsigned nb = (unsigned)(((D->GL.Nj + 7) / 8) * ((D->GL.Nk + 7) / 8));
    const int wave_on = env_int("KRYST_ILU_WAVE", default_wave_form(nb));
    const bool quad = wave_on >= 2 && D->GL.d_blocked && D->GU.d_blocked;
    info[0] = D->safe ? 3 : quad ? 2 : 1;
    info[1] = D->GL.Ni; info[2] = D->GL.Nj; info[3] = D->GL.Nk;
    if (quad) {
        KR_HIP(hipSetDevice(ctx->device));
        int q = 0;
        for (GridFactor* G : {&D->GL, &D->GU}) {
            const size_t cnt = (size_t)G->nbj * G->nbk * 4 * (size_t)G->nch;
            info[6 + q] = (int64_t)cnt;
            info[10 + q] = (int64_t)((q == 0 ? 3 : 4) * 4 * 64 * sizeof(tw_v2));
            if (G->d_skip && cnt) {
                std::vector<uint8_t> h(cnt);
                KR_HIP(hipMemcpyAsync(h.data(), G->d_skip, cnt, hipMemcpyDeviceToHost, ctx->s_main));
                KR_HIP(hipStreamSynchronize(ctx->s_main));
                int64_t ns = 0;
                for (uint8_t v : h) ns += v;
                info[8 + q] = ns;
            }
            ++q;
        }
    }
    return KRYST_OK;
}

#ifdef KR_TW_TRACE
extern "C" int32_t kryst_debug_tq_trace(long long* host, int32_t count) {
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(kr::tq_trace), sizeof(long long) * count) == hipSuccess ? 0 : 1;
}
extern "C" int32_t kryst_debug_tq_select(int32_t a, int32_t b) {
    const int v[2] = {a, b};
    return hipMemcpyToSymbol(HIP_SYMBOL(kr::tq_sel), v, sizeof(v)) == hipSuccess ? 0 : 1;
}
extern "C" int32_t kryst_debug_tq_steps(long long* host) {          // 3 x 1024: solver steps, poller deliveries, exporter stores
    if (hipMemcpyFromSymbol(host, HIP_SYMBOL(kr::tq_steps), sizeof(long long) * 1024) != hipSuccess) return 1;
    if (hipMemcpyFromSymbol(host + 1024, HIP_SYMBOL(kr::tq_deliv), sizeof(long long) * 1024) != hipSuccess) return 1;
    return hipMemcpyFromSymbol(host + 2048, HIP_SYMBOL(kr::tq_export), sizeof(long long) * 1024) == hipSuccess ? 0 : 1;
}
extern "C" int32_t kryst_debug_tw_trace(long long* host, int32_t count) {
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(kr::tw_trace), sizeof(long long) * count) == hipSuccess ? 0 : 1;
}
extern "C" int32_t kryst_debug_tw_steps(long long* host, int32_t count) {
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(kr::tw_steps), sizeof(long long) * count) == hipSuccess ? 0 : 1;
}
extern "C" int32_t kryst_debug_tw_rounds(long long* host, int32_t count) {
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(kr::tw_rounds), sizeof(long long) * count) == hipSuccess ? 0 : 1;
}
#endif
