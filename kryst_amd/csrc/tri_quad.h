// PIPELINED WAVEFRONT solve of a structured-grid factor, third generation: 16 x 16 grid lines per workgroup.
//
// tri_wave.h solves an 8 x 8 block of lines with ONE solving wave per workgroup and is bound, at 256^3, by (i) the block hops of
// the pipeline fill -- 62 of them, each a trip through memory and a poller round -- and (ii) the chip holding only one such
// workgroup per CU (its loader wave streams five arrays through 284 registers), i.e. 64 lines per CU.  Here a workgroup owns a
// 16 x 16 block as FOUR 8 x 8 quadrants, one solving wave each, and
//   * neighbouring quadrants hand their edge rows over through rings in LDS (~0.1 us) -- only the 16 + 16 lines on the
//     workgroup's west and south boundary come through memory, so a 256^3 solve has 30 memory hops instead of 62 and every
//     block of the grid is resident at once (256 workgroups);
//   * the factor's coefficients are stored at setup in BLOCKED layout -- [block][chunk of 8 steps][array][step pair][lane],
//     zero where a lane is outside its line -- so a solving wave loads them itself, 1 KiB per instruction, two chunks ahead
//     (inline-asm requests counted by hand), and needs no activity predicate in its step (a zero coefficient selects a +0.0
//     operand, as in tri_wave.h); a chunk whose coefficients repeat chunk - 3's bit for bit (flags from setup: the factors of
//     constant-coefficient operators settle to the last bit a few rows from the box's low faces) is not requested at all;
//   * the right-hand side, which lives in the caller's natural layout, goes through two LOADER waves and an LDS stage; the
//     same waves store the RESULTS, read out of the solving waves' rings as whole 64-byte groups, four lanes per line -- a
//     solving wave issues no global store and, away from the low faces, no load;
//   * what a neighbouring workgroup needs is written by an EXPORTER wave to compact EDGE buffers -- [block][producer step]
//     [16 lines], one memory line per step -- instead of being picked out of the solution vector: a POLLER round is two
//     contiguous 1 KiB reads.
// Lane (jl, kl) of a quadrant walks its line one row per step, skewed by jl + kl; quadrants run on their own clocks: a consumer
// at step t needs its producer (sibling quadrant or neighbouring workgroup) to have finished step t + 7.
// Same arithmetic and order as every other triangular-solve form (stored, i.e. ascending-column, order): bit-identical.
//
// Progress relies on the blocks a resident block waits for being resident or finished (workgroups start in index order, blocks
// are numbered along anti-diagonals): an observed property, not a HIP guarantee -- every spin has a budget, and a poller that
// runs out of patience raises the host-visible give-up word (ilu.hip: ilu_health -> plane kernels).
#pragma once

namespace kr {

// -DTQ_ABL=bits: TIMING-ONLY ablations (wrong results; never in the shipped library): 1 no result stores, 2 coefficients from four
// cached chunks, 4 spins sleep 4x longer, 8 no west/south ring reads, 16 no lane exchange, 32 no ring write, 64 no stage read,
// 128 a poller stream counts as fully delivered once its first rows have arrived, 256 the solving waves never wait for the result stores,
// 512 every result store goes into the vector's first 32 KiB.
#if defined(TQ_ABL) && (TQ_ABL & 4)
#define TQ_NAP(n) __builtin_amdgcn_s_sleep(4 * (n))
#else
#define TQ_NAP(n) __builtin_amdgcn_s_sleep(n)
#endif
constexpr int TQ_C = 8;            // steps per chunk
constexpr int TQ_S = 4;            // LDS stage slots (right-hand side) -- per loader half, whose two quadrants run 8+ steps apart
constexpr int TQ_YR = 32;          // steps in a solving wave's result ring
constexpr int TQ_R = 64;           // steps in a neighbour ring
constexpr int TQ_LINES = 256;      // lines per workgroup (4 quadrants x 64 lanes)

struct QuadView {
    int32_t Ni, Nj, Nk, nbj, nbk, nch;
    const tw_v2* coef;             // blocked coefficients: [block][chunk][array][step pair][line] (array: i-, j-, k-neighbour[, divisor])
    double* edge_e; double* edge_n;   // [block][nch * 8 + 8 producer steps][16 lines]
    const uint8_t* skip;           // [block][quadrant][chunk]: this chunk's coefficients equal chunk - 3's bit for bit (nullptr: no flags)
};
constexpr int TQ_SKIPMAX = 544;    // chunks per line the flag table in LDS holds (lines up to 4 300 rows); longer lines run without flags

// Counters in LDS that tie the waves of a workgroup together.  The payload they guard is in LDS as well, and the LDS executes
// one wave's operations in issue order, so publishing needs NO wait at all: the ring write and the counter write that follows it
// reach the LDS in that order, and a reader that has seen the counter issues its ring reads afterwards.  Only the compiler must
// keep the program order (the empty asm).  The workgroup-scope release / acquire forms would wait for lgkmcnt(0) -- one LDS
// round trip per publish on the solving wave's critical path -- and for vmcnt(0): every outstanding GLOBAL load and store of
// the wave, i.e. an HBM round trip per step for a wave that keeps a chunk of coefficient loads in flight.
__device__ __forceinline__ double tq_shr1(double v) {                      // lane l <- lane l - 1 within its row of 16 lanes (row_shr:1)
    // (bound_ctrl: a lane without a source -- lane 0 of a row -- reads 0 instead of keeping `old`: no register to initialise; those lanes
    // take the west operand anyway)
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x111, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x111, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ void tq_publish(int* p, int v) {
    asm volatile("" ::: "memory");
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    asm volatile("" ::: "memory");
}
__device__ __forceinline__ int tq_peek(int* p) {
    asm volatile("" ::: "memory");
    const int v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    asm volatile("" ::: "memory");
    return v;
}

#ifdef KR_TW_TRACE
// trace build (tools/tq_trace.py): per block and quadrant of the FORWARD solve: [0] entry, [1] first coefficients in registers,
// [2 + c] chunk c done (c < 8), [10] last chunk done, [11] ticks waiting for neighbour rows, [12] for the stage, [13] for sibling ring space
__device__ long long tq_trace[4096 * 4 * 16];
// step-level trace of two chosen blocks (tq_sel, linear K * nbj + J): [which][quadrant or stream][step < 128]
__device__ int tq_sel[2] = {-1, -1};
__device__ long long tq_steps[2 * 4 * 128], tq_deliv[2 * 4 * 128], tq_export[2 * 4 * 128];
#define TQ_STEP_TRACE(arr, qq, step_) do { if (FORWARD && tq_which >= 0 && (step_) < 128) arr[(tq_which * 4 + (qq)) * 128 + (step_)] = wall_clock64(); } while (0)
#define TQ_STAMP(slot) do { if (FORWARD && l == 0 && blk < 4096) tq_trace[(blk * 4 + q) * 16 + (slot)] = wall_clock64(); } while (0)
#define TQ_T0(name) const long long name = wall_clock64()
#define TQ_ACC(slot, t0_) do { tq_acc[(slot) - 11] += wall_clock64() - (t0_); } while (0)      // (registers: a read-modify-write of memory would stall the wave)
#else
#define TQ_STAMP(slot) do { } while (0)
#define TQ_ACC(slot, t0_) do { } while (0)
#define TQ_T0(name) do { } while (0)
#define TQ_STEP_TRACE(arr, qq, step_) do { } while (0)
#endif

__device__ __forceinline__ void tq_block_of(int b, int nbj, int nbk, int& J, int& K) {     // anti-diagonal numbering
    int d = 0, rem = b, lo = 0;
    for (;; ++d) {
        lo = max(0, d - (nbk - 1));
        const int cnt = min(d, nbj - 1) - lo + 1;
        if (rem < cnt) break;
        rem -= cnt;
    }
    J = lo + rem; K = d - J;
}

// line index L (0..255) of a workgroup: quadrant q = L >> 6 (qj = q & 1, qk = q >> 1), lane l8 = L & 63 (jl8 = l8 & 7, kl8 = l8 >> 3)
struct TqLine { int jj, kk, skew; };
__device__ __forceinline__ TqLine tq_line(int J, int K, int L) {
    const int q = L >> 6, l8 = L & 63;
    return TqLine{16 * J + 8 * (q & 1) + (l8 & 7), 16 * K + 8 * (q >> 1) + (l8 >> 3), (l8 & 7) + (l8 >> 3)};
}

// ---- setup: natural-order coefficient streams -> blocked layout.  One workgroup of 256 threads per (block, chunk).
template <bool FORWARD, int NA>
__global__ __launch_bounds__(256) void tri_quad_layout_kernel(GridView G, int nbj, int nbk, int nch, tw_v2* coef) {
    const int blk_lin = blockIdx.x / nch, c = blockIdx.x % nch, L = threadIdx.x;
    const int J = blk_lin % nbj, K = blk_lin / nbj;                        // (layout is indexed by K * nbj + J)
    const TqLine ln = tq_line(J, K, L);
    const bool line_ok = ln.jj < G.Nj && ln.kk < G.Nk;
    const int j = FORWARD ? ln.jj : G.Nj - 1 - ln.jj, k = FORWARD ? ln.kk : G.Nk - 1 - ln.kk;
    const int64_t line0 = line_ok ? (int64_t)(k * G.Nj + j) * G.Ni : 0;
    const double* src[4] = {G.c1, G.c2, G.c3, G.diag};
    tw_v2* dst = coef + ((size_t)blk_lin * nch + c) * NA * 4 * TQ_LINES + L;
#pragma unroll
    for (int a = 0; a < NA; ++a)
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            double v[2];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int ii = c * TQ_C + 2 * h + e - ln.skew;
                const bool ok = line_ok && ii >= 0 && ii < G.Ni;
                const int64_t row = line0 + (FORWARD ? ii : G.Ni - 1 - ii);
                v[e] = ok ? src[a][row] : (a == 3 ? 1.0 : 0.0);           // outside the line: no entries, divisor 1
            }
            dst[(a * 4 + h) * TQ_LINES] = tw_v2{v[0], v[1]};
        }
}

// ---- setup: which coefficient chunks repeat.  A solving wave keeps chunk m in register buffer m % 3, so a chunk that equals chunk
// m - 3 bit for bit is ALREADY there and needs no request.  The ILU factors of constant-coefficient operators settle to the last
// bit 10-40 rows away from the low faces of the box (the BASELINE operators: 10-15 rows), so away from those faces the solve
// streams only the right-hand side and the result: 32 instead of 88 bytes per row and apply.  Exact: equal bits, same arithmetic.
// One workgroup per (block, chunk); a quadrant's 64 lines are one wave.
template <int NA>
__global__ __launch_bounds__(256) void tri_quad_dedup_kernel(const tw_v2* coef, int nch, uint8_t* skip) {
    const int blk = blockIdx.x / nch, c = blockIdx.x % nch, L = threadIdx.x, q = L >> 6;
    bool eq = c >= 3;
    if (c >= 3) {
        const unsigned long long* cur = reinterpret_cast<const unsigned long long*>(coef + ((size_t)blk * nch + c) * NA * 4 * TQ_LINES + L);
        const unsigned long long* old = reinterpret_cast<const unsigned long long*>(coef + ((size_t)blk * nch + c - 3) * NA * 4 * TQ_LINES + L);
#pragma unroll
        for (int e = 0; e < NA * 4; ++e)
            eq = eq && cur[2 * e * TQ_LINES] == old[2 * e * TQ_LINES] && cur[2 * e * TQ_LINES + 1] == old[2 * e * TQ_LINES + 1];
    }
    const bool all = __all(eq);                                            // (bit patterns, so that -0.0 / NaN payloads count)
    if ((L & 63) == 0) skip[((size_t)blk * 4 + q) * nch + c] = all ? 1 : 0;
}

// ---- ONCE, at setup: sentinels into the edge buffers (zeros in the 8 steps past the last chunk: rows nobody has), flags and abort
// word cleared.  Between applies the pollers re-arm the cells they consumed and the flags carry the apply's number.
__global__ __launch_bounds__(256) void tri_quad_fill_kernel(double* edge_e, double* edge_n, int nblk, int nch, int32_t* flags, int32_t nflags) {
    const int64_t per = (int64_t)(nch * TQ_C + 8) * 16;                   // doubles per block and direction
    const int64_t total = per * nblk;
    const double sentinel = __longlong_as_double((long long)KR_TRI_SENTINEL);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const double v = (i % per) < (int64_t)nch * TQ_C * 16 ? sentinel : 0.0;
        edge_e[i] = v; edge_n[i] = v;
    }
    for (int i = blockIdx.x * 256 + threadIdx.x; i < nflags; i += gridDim.x * 256) flags[i] = 0;
}

// An apply launched directly (not replayed from a graph) gets the caller's vectors, the solver's `done` flag and its number as KERNEL
// ARGUMENTS (epoch != 0) instead of through the device argument block: no argument kernel, one kernel boundary less per apply.
// Direct epochs live in [2^30, 2^31), the argument block's in [1, 2^30): the two never meet in a flag.
struct TriDirect { const double* r; double* z; const int* done; int32_t epoch; };
template <bool FORWARD>
__global__ __launch_bounds__(512, 2) void tri_quad_kernel(const TriArgs* args, const double* in_ptr, double* out_ptr, QuadView Q, int64_t n,
                                                          int32_t* flags, int32_t* abort_word, int32_t* gave_up, int poll_budget, const TriDirect dir) {
    if (dir.epoch != 0 ? (dir.done && *dir.done) : (args->skip != 0)) return;
    const int32_t epoch = dir.epoch != 0 ? dir.epoch : (int32_t)args->epoch;   // this apply's number: the value of an "under way" flag (nothing resets the flags)
    constexpr int C = TQ_C, S = TQ_S, R = TQ_R;
    constexpr int NA = FORWARD ? 3 : 4;                                    // coefficient arrays per chunk
    constexpr int YR = TQ_YR;
    __shared__ __attribute__((aligned(16))) double stage[S * C * TQ_LINES];   // right-hand side: [slot][step pair][line][2]: one 16-byte read per two steps   64 KiB
    __shared__ double yring[4 * YR * 64];                                  // every solving wave's last YR steps, all 64 lanes             64 KiB
    __shared__ double pring[4 * R * 8];                                    // poller streams (west of q0, west of q2, south of q0, south of q1): [step % R][edge lane]  16 KiB
    __shared__ int prog[4];                                                // steps each solving wave has finished (its rows are in its ring)
    __shared__ int pavail[4];                                              // consumer steps each poller stream has delivered
    __shared__ int taken[4];                                               // chunks each solving wave has taken off the stage
    __shared__ int exported[4];                                            // producer steps each exporter stream has written to the edge buffers
    __shared__ int written[4];                                             // chunks of each solving wave's results the loader waves have stored to the caller's vector
    __shared__ uint8_t skipf[4 * TQ_SKIPMAX];                              // per solving wave and chunk m, a 3-bit code: bit d = chunk m + d needs a coefficient request
    __shared__ int staged[2], quit, always, gate;                          // gate: the producers are under way (set by the poller)                                // chunks each loader wave has staged (quadrants 0-1 / 2-3)
    cgdouble* in = (cgdouble*)(in_ptr ? in_ptr : (dir.epoch != 0 ? dir.r : args->r));
    gdouble* out = (gdouble*)(out_ptr ? out_ptr : (dir.epoch != 0 ? dir.z : args->z));
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);    // 0-3 solve, 4-5 load the right-hand side (128 lines each), 6 polls, 7 exports
    const int l = threadIdx.x & 63;
    int J, K;
    tq_block_of(blockIdx.x, Q.nbj, Q.nbk, J, K);
    const int blk = K * Q.nbj + J;
#ifdef KR_TW_TRACE
    const int tq_which = blk == tq_sel[0] ? 0 : blk == tq_sel[1] ? 1 : -1;   // (read once: a load of the selection per step would pace the kernel)
#endif
    const int nch = Q.nch, T = nch * C;
    const int64_t edge_stride = (int64_t)(T + 8) * 16;                    // doubles per block in an edge buffer
    constexpr int HUGE_STEPS = 1 << 30;
    
    if (threadIdx.x < 4) {
        const int i = threadIdx.x;
        prog[i] = 0; taken[i] = 0; pavail[i] = 0; written[i] = 0;
        // exporter streams: 0 east rows of q1, 1 east rows of q3, 2 north rows of q2, 3 north rows of q3
        exported[i] = (i < 2 ? J + 1 < Q.nbj : K + 1 < Q.nbk) ? 0 : HUGE_STEPS;
    }
    if (threadIdx.x == 0) { staged[0] = 0; staged[1] = 0; quit = 0; always = HUGE_STEPS; gate = (J == 0 && K == 0) ? 1 : 0; }
    for (int i = threadIdx.x; i < 4 * TQ_SKIPMAX; i += 512) {               // (before any hand-counted request is in flight)
        const int qq = i / TQ_SKIPMAX, m = i % TQ_SKIPMAX;
        // (one look-up per chunk answers all three questions the solving wave has about it and the two requests behind it: four
        // separate flag reads, each a dependent LDS round trip on the wave's critical path, cost 8 % of a chunk)
        const bool flags_on = Q.skip && nch + 3 <= TQ_SKIPMAX;
        unsigned code = 0;
        for (int d = 0; d < 3; ++d) {
            const int md = m + d;
            const bool need = md < nch && !(flags_on && Q.skip[((size_t)blk * 4 + qq) * nch + md] != 0);
            code |= need ? (1u << d) : 0u;
        }
        skipf[i] = (uint8_t)code;
    }
    for (int i = threadIdx.x; i < 4 * R * 8; i += 512) pring[i] = 0.0;
    for (int i = threadIdx.x; i < 4 * YR * 64; i += 512) yring[i] = 0.0;
    __syncthreads();                                                      // the only barrier

    // ------------------------------------------------------------------------------------------------ the LOADER
    if (wave == 4 || wave == 5) {
        const int half = wave - 4;                                        // lines 128 half .. 128 half + 127 (quadrants 2 half, 2 half + 1)
        // right-hand side, natural layout -> stage.  Interior chunks: four lanes share a line's 64 contiguous bytes (lane 4g + c
        // of pass r loads the 16-byte piece c of line 16 r + g), so one load instruction touches 16 memory lines, not 64.
        const int g = l >> 2, c = l & 3;
        const bool full = 16 * J + 16 <= Q.Nj && 16 * K + 16 <= Q.Nk;
        auto fast_chunk = [&](int t0) { return full && t0 >= 14 && t0 + C <= Q.Ni; };     // every line of the block inside [0, Ni) for all 8 steps
        // geometry of this lane's eight lines (16 r + g of the half), once: first row, skew, inside the grid?  (Recomputing it per
        // load -- a division-free but 64-bit multiply-heavy line lookup, eight times per chunk in both halves of the careful path --
        // made the loader's start-up 0.7 us longer per block, and the loader's first chunk is what a block's first step waits for.)
        int64_t ln0[8]; int lsk[8]; bool lok[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const TqLine ln = tq_line(J, K, 128 * half + 16 * r + g);
            lok[r] = ln.jj < Q.Nj && ln.kk < Q.Nk;
            const int jx = FORWARD ? ln.jj : Q.Nj - 1 - ln.jj, kx = FORWARD ? ln.kk : Q.Nk - 1 - ln.kk;
            ln0[r] = lok[r] ? (int64_t)(kx * Q.Nj + jx) * Q.Ni : 0;
            lsk[r] = ln.skew;
        }
        auto row0 = [&](int r) -> int64_t { return ln0[r] + (FORWARD ? -lsk[r] : Q.Ni - 1 + lsk[r]); };   // row of step 0 of line 16 r + g (may lie outside the line)
        // The loads are issued and waited for BY HAND (inline asm + s_waitcnt with the exact number of younger loads): three
        // chunks are in flight and the two paths below differ, and the compiler's automatic vmcnt then assumes the worst at every
        // merge and drains all buffers at each publish -- one HBM round trip per chunk, which was the whole kernel's pace.
        struct Buf { tw_v2 d[8]; };
        // careful path (lines start / end inside the chunk, ragged blocks): the SAME lane -> (line, 16-byte piece) mapping as the
        // interior chunks, so that the block's first and last chunks -- which every hand-over to the next block waits for -- are
        // as coalesced as the others (two whole lines per lane, 64 memory lines per instruction, cost 5-8 % of a 256^3 apply); a
        // step pair that sticks out of its line by one row is loaded one row further in and the missing half replaced by zero
        // when the buffer is published, a pair wholly outside is loaded from the line's first rows and published as zeros
        auto slow_pair = [&](int t0, int r, int h, bool& oka, bool& okb, int64_t& lo) {
            const int ia = t0 + 2 * h - lsk[r], ib = ia + 1;                 // line coordinates of the pair's two steps
            oka = lok[r] && ia >= 0 && ia < Q.Ni; okb = lok[r] && ib >= 0 && ib < Q.Ni;
            const int w = (oka && !okb) ? ia - 1 : (!oka && okb) ? ib : ia;  // a 2-row window inside the line (Ni >= 2)
            const int wc = (oka || okb) ? w : 0;
            lo = ln0[r] + (FORWARD ? wc : Q.Ni - 2 - wc);                    // lower memory row of the window
        };
        auto fetch = [&](Buf& b, int t0) {
            if (fast_chunk(t0)) {
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    const int64_t lo = FORWARD ? row0(r) + t0 : row0(r) - t0 - (C - 1);       // lowest row of the chunk
                    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(b.d[r]) : "v"(in + lo + 2 * c) : "memory");
                }
            } else {
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    bool oka, okb; int64_t lo;
                    slow_pair(t0, r, c, oka, okb, lo);
                    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(b.d[r]) : "v"(in + lo) : "memory");
                }
            }
        };
        // `younger`: loads issued after this buffer's eight (0, 8 or 16)
        auto publish = [&](Buf& b, int kc, int younger) {
            // slot kc % S is free once every solving wave has taken chunk kc - S
            for (int budget = 1 << 24; budget > 0; --budget) {
                const int m = min(tq_peek(&taken[2 * half]), tq_peek(&taken[2 * half + 1]));
                if (kc - m < S || tq_peek(&quit)) break;
                TQ_NAP(2);
            }
            if (younger >= 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
            else if (younger >= 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int r = 0; r < 8; ++r) asm volatile("" : "+v"(b.d[r]));     // (the values exist from here on)
            double* dst = stage + (size_t)(kc % S) * C * TQ_LINES;
            if (fast_chunk(kc * C)) {
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    const int L = 128 * half + 16 * r + g;
                    // the piece's two rows are steps 2c, 2c + 1 (forward) or 7 - 2c, 6 - 2c (backward): one step pair either way
                    const int pr = FORWARD ? c : 3 - c;
                    *reinterpret_cast<tw_v2*>(dst + (pr * TQ_LINES + L) * 2) = FORWARD ? b.d[r] : tw_v2{b.d[r].y, b.d[r].x};
                }
            } else {
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    bool oka, okb; int64_t lo;
                    slow_pair(kc * C, r, c, oka, okb, lo);
                    const tw_v2 pr = b.d[r];
                    const double ra = FORWARD ? pr.x : pr.y, rb = FORWARD ? pr.y : pr.x;      // line rows wc, wc + 1
                    const double va = (oka && !okb) ? rb : ra, vb = (!oka && okb) ? ra : rb;
                    *reinterpret_cast<tw_v2*>(dst + (c * TQ_LINES + 128 * half + 16 * r + g) * 2) = tw_v2{oka ? va : 0.0, okb ? vb : 0.0};   // steps 2c, 2c + 1
                }
            }
            tq_publish(&staged[half], kc + 1);
        };
        // ---- RESULT stores of this half's two solving waves (so that a solving wave issues no store at all: 0.3 us of its 2 us
        // per chunk, on the critical path of every hand-over).  A line's rows leave in aligned groups of 8 (64 bytes), four lanes
        // per line -- one store instruction writes 16 whole 64-byte segments -- read out of the solving wave's ring: line L's
        // group m = kc - ceil(skew_L / 8) is complete once the wave has finished chunk kc (taken[q] > kc).  (Each lane storing
        // 16 bytes of ITS line, 64 lines per instruction and four partial writes per segment, cost 1.5 ms of a 4.2 ms apply at
        // 512^3.)  Chunk nch is the tail: groups that were not complete two chunks before the end.  First / last chunks and
        // ragged blocks: lane = line, row by row.
        const int sg = l >> 2, sc = l & 3;                                  // store r of a chunk: line 16 r + sg of the quadrant, 16-byte piece sc
        const int g_skew0 = (sg & 7) + (sg >> 3);                           // skew of that line: g_skew0 + 2 r
        // (geometry once: line 16 r + sg of quadrant 2 half + qd starts at w_l0 + qd * w_dq + r * w_dr)
        const int w_jj0 = 16 * J + (sg & 7), w_kk0 = 16 * K + 8 * half + (sg >> 3);          // quadrant qd adds 8 to jj, store r adds 2 to kk
        const int64_t w_l0 = (int64_t)((FORWARD ? w_kk0 : Q.Nk - 1 - w_kk0) * Q.Nj + (FORWARD ? w_jj0 : Q.Nj - 1 - w_jj0)) * Q.Ni;
        const int64_t w_dq = (FORWARD ? 8 : -8) * (int64_t)Q.Ni, w_dr = (FORWARD ? 2 : -2) * (int64_t)Q.Nj * Q.Ni;
        auto write_chunk = [&](int qq, int kc) {
            const int qd = qq & 1;
            const double* const ring_q = yring + qq * YR * 64;
            const int64_t l00 = w_l0 + qd * w_dq;
            if (full && kc >= 2 && 8 * kc + 8 <= Q.Ni) {
#if defined(TQ_ABL) && (TQ_ABL & 1)
                return;
#endif
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int L = 16 * r + sg, gs = g_skew0 + 2 * r, m = kc - ((gs + 7) >> 3);
                    const int ia = FORWARD ? 8 * m + 2 * sc : 8 * m + 7 - 2 * sc, ib = FORWARD ? ia + 1 : ia - 1;   // line rows of the piece, in memory order
                    tw_v2 v;
                    v.x = ring_q[((ia + gs) & (YR - 1)) * 64 + L]; v.y = ring_q[((ib + gs) & (YR - 1)) * 64 + L];
#if defined(TQ_ABL) && (TQ_ABL & 512)
                    const int64_t lo = (l00 + r * w_dr + (FORWARD ? 8 * m : Q.Ni - 8 - 8 * m)) & 4095;              // timing only: every result store into the vector's first 32 KiB (cache hits)
#else
                    const int64_t lo = l00 + r * w_dr + (FORWARD ? 8 * m : Q.Ni - 8 - 8 * m);                       // lowest memory row of the group
#endif
                    *(__attribute__((address_space(1))) tw_v2*)(out + lo + 2 * sc) = v;
                }
            } else {
                // first / last chunks, ragged blocks: the same four-lanes-per-line mapping, every piece with its own predicate (a
                // group not complete yet has m < 0; a piece that straddles the line's end stores its one row)
                const bool j_ok = w_jj0 + 8 * qd < Q.Nj;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int L = 16 * r + sg, gs = g_skew0 + 2 * r, m = kc - ((gs + 7) >> 3);
                    const bool line_ok = j_ok && w_kk0 + 2 * r < Q.Nk;
                    const int ia = FORWARD ? 8 * m + 2 * sc : 8 * m + 7 - 2 * sc, ib = FORWARD ? ia + 1 : ia - 1;   // line rows of the piece, in memory order
                    const bool oka = line_ok && m >= 0 && ia >= 0 && ia < Q.Ni, okb = line_ok && m >= 0 && ib >= 0 && ib < Q.Ni;
                    const double ya = ring_q[((ia + gs) & (YR - 1)) * 64 + L], yb = ring_q[((ib + gs) & (YR - 1)) * 64 + L];
                    gdouble* const at = out + (line_ok ? l00 + r * w_dr : 0) + (FORWARD ? ia : Q.Ni - 1 - ia);     // memory row of line row ia; ib is the next one
                    if (oka && okb) *(__attribute__((address_space(1))) tw_v2*)at = tw_v2{ya, yb};
                    else if (oka) at[0] = ya;
                    else if (okb) at[1] = yb;
                }
            }
        };
        int wdone0 = 0, wdone1 = 0;                                        // chunks written per quadrant of this half (0 .. nch + 1: the tail counts as one)
        auto drain = [&]() {                                               // everything the solving waves have finished since the last call
#pragma unroll 1
            for (int qd = 0; qd < 2; ++qd) {
                const int qq = 2 * half + qd;
                const int have = tq_peek(&taken[qq]);                      // chunks whose steps are done (their rows are in the ring)
                int wd = qd ? wdone1 : wdone0;
#pragma unroll 1
                while (wd < have || (wd == nch && have >= nch)) {          // (chunk nch: the tail)
                    write_chunk(qq, wd);
                    ++wd;
                    if (l == 0 && wd <= nch) tq_publish(&written[qq], wd); // (same-wave LDS reads above are done: the ring slots are free)
                }
                if (qd) wdone1 = wd; else wdone0 = wd;
            }
        };
        // nothing is requested before the blocks this one depends on are under way: a block that will not run for another
        // 100 us must not queue its first chunks in front of the blocks at the front
        for (int budget = 1 << 24; tq_peek(&gate) == 0 && budget > 0; --budget) TQ_NAP(8);
        Buf b0, b1, b2;                                                    // chunk kc lives in buffer kc % 3
        fetch(b0, 0);
        if (1 < nch) fetch(b1, C);
        if (2 < nch) fetch(b2, 2 * C);
        // Static buffer roles (the loop body is three chunks long): the requests are inline asm the register allocator knows
        // nothing about, and any other loop shape (one chunk per trip, buffer chosen by a switch) made it COPY the buffers between
        // trips while their loads were in flight.  `younger` of a publish: the loads of the chunks kc + 1 .. that have been
        // requested by now (result stores issued in between only make the count conservative).
        for (int kc = 0; kc < nch; kc += 3) {
            publish(b0, kc, 8 * (min(nch, kc + 3) - (kc + 1)));
            if (kc + 3 < nch) fetch(b0, (kc + 3) * C);
            drain();
            if (kc + 1 < nch) { publish(b1, kc + 1, 8 * (min(nch, kc + 4) - (kc + 2))); if (kc + 4 < nch) fetch(b1, (kc + 4) * C); drain(); }
            if (kc + 2 < nch) { publish(b2, kc + 2, 8 * (min(nch, kc + 5) - (kc + 3))); if (kc + 5 < nch) fetch(b2, (kc + 5) * C); drain(); }
        }
#pragma unroll 1
        for (int budget = 1 << 26; budget > 0; --budget) {                 // the solving waves' last chunks and the tails
            drain();
            if ((wdone0 > nch && wdone1 > nch) || tq_peek(&quit)) break;
            TQ_NAP(1);
        }
        return;
    }

    // ------------------------------------------------------------------------------------------------ the POLLER
    if (wave == 6) {
        // four streams of 16 lanes each: g = 0 / 1 west inputs of quadrants (0,0) / (0,1), g = 2 / 3 south inputs of quadrants
        // (0,0) / (1,0).  Lane idx of a stream: step offset idx >> 2 (and + 4 for the second load), 16-byte piece idx & 3 = edge
        // lanes 2 piece, 2 piece + 1.  Consumer step t of a stream reads the producer's step t + 7.
        const int gI = l >> 4, idx = l & 15, so = idx >> 2, piece = idx & 3;
        const bool west = gI < 2;
        const int cq = gI == 0 ? 0 : gI == 1 ? 2 : gI == 2 ? 0 : 1;      // consumer quadrant (q = qj + 2 qk)
        const bool has_src = west ? J > 0 : K > 0;
        const int src_blk = west ? blk - 1 : blk - Q.nbj;
        gdouble* const src = (gdouble*)(west ? Q.edge_e : Q.edge_n) + (int64_t)(has_src ? src_blk : blk) * edge_stride + 8 * (gI & 1) + 2 * piece;
        double* const ring = pring + gI * R * 8;
        int* const avail = &pavail[gI];
        // GATE: until the producers are under way, look at their flags only
        if (l == 0) {
            for (int budget = 1 << 22; budget > 0; --budget) {
                const bool ok_w = J == 0 || __hip_atomic_load(&flags[blk - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == epoch;
                const bool ok_s = K == 0 || __hip_atomic_load(&flags[blk - Q.nbj], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == epoch;
                if (ok_w && ok_s) break;
                if ((budget & 63) == 0 && __hip_atomic_load(abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) break;
                TQ_NAP(8);
            }
            tq_publish(&gate, 1);
        }
        int t = has_src ? 0 : T;                                           // next consumer step of this lane's stream
        for (int budget = poll_budget; budget > 0;) {
            if (__ballot(t < T) == 0) break;
            const int free_upto = max(tq_peek(&taken[cq]) - 1, 0) * C + R;      // ring slots below this consumer step are free
            const bool go = t < T && t + 8 <= free_upto;
            if (__ballot(go) == 0) { TQ_NAP(2); --budget; continue; }
            gdouble* const a0 = go ? src + (int64_t)(t + 7 + so) * 16 : (gdouble*)Q.edge_e + (int64_t)blk * edge_stride;
            gdouble* const a1 = go ? a0 + 4 * 16 : a0;
            tw_v2 p0, p1;                                                 // agent-scope (sc1) 16-byte loads: each 8-byte half whole or sentinel
            asm volatile("global_load_dwordx4 %0, %2, off sc1\n\tglobal_load_dwordx4 %1, %3, off sc1\n\ts_waitcnt vmcnt(0)"
                         : "=&v"(p0), "=&v"(p1) : "v"(a0), "v"(a1) : "memory");
            const bool bad0 = go && (tw_is_sentinel(p0.x) || tw_is_sentinel(p0.y)), bad1 = go && (tw_is_sentinel(p1.x) || tw_is_sentinel(p1.y));
            const unsigned long long b0m = __ballot(bad0), b1m = __ballot(bad1);
            // leading complete steps of this lane's stream: step offset s of load 0 = lanes 4 s .. 4 s + 3 of the stream
            const unsigned s0 = (unsigned)(b0m >> (16 * gI)) & 0xffffu, s1 = (unsigned)(b1m >> (16 * gI)) & 0xffffu;
            int m = 0;
#pragma unroll
            for (int sidx = 0; sidx < 8; ++sidx) {
                const unsigned bits = sidx < 4 ? (s0 >> (4 * sidx)) & 0xfu : (s1 >> (4 * (sidx - 4))) & 0xfu;
                if (m == sidx && bits == 0) m = sidx + 1;
            }
            if (!go) m = 0;
            const unsigned long long stuck = __ballot(go && m == 0);
            if (stuck && (budget == 1 || ((budget & 255) == 0 && __hip_atomic_load(abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0))) {
                // out of patience (tri_wave.h: same rule): hand over what there is, tell every block, raise the host-visible word
                __hip_atomic_store(abort_word, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(gave_up, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                tq_publish(&quit, 1);
                if (l < 4) { tq_publish(&pavail[l], HUGE_STEPS); tq_publish(&prog[l], HUGE_STEPS); tq_publish(&exported[l], HUGE_STEPS); }
                return;
            }
            // delivered rows go into the ring (their cells in the edge buffer go back to the sentinel after the loop)
            if (so < m) { ring[((t + so) % R) * 8 + 2 * piece] = p0.x; ring[((t + so) % R) * 8 + 2 * piece + 1] = p0.y; }
            if (so + 4 < m) { ring[((t + so + 4) % R) * 8 + 2 * piece] = p1.x; ring[((t + so + 4) % R) * 8 + 2 * piece + 1] = p1.y; }
            if (m > 0) { if (idx == 0) for (int z = 0; z < m; ++z) TQ_STEP_TRACE(tq_deliv, gI, t + z); t += m; if (idx == 0) tq_publish(avail, t); }
#if defined(TQ_ABL) && (TQ_ABL & 128)
            // timing only: once a stream's FIRST rows have arrived, everything counts as delivered (what the per-step coupling to the
            // producers costs on top of the start-up hand-over)
            if (t > 0 && t < T) { t = T; if (idx == 0) tq_publish(avail, T); }
#endif
#ifdef TQ_SLOWSTART
            if (stuck) { if (__ballot(t > 0) == 0) TQ_NAP(TQ_SLOWSTART); else TQ_NAP(1); --budget; }
#else
            if (stuck) { TQ_NAP(1); --budget; }
#endif
        }
        // RE-ARM, once, when everything has been delivered: the consumed cells (producer steps 7 .. T - 1 of this stream's eight edge
        // lanes) go back to the sentinel for the NEXT apply (this block is their only reader; the kernel boundary orders the stores
        // before the producer's next write) -- so that no launch has to re-arm the buffers between applies.  Rows past the producer's
        // last step are zeros for good.  (Round 2 did it inside the loop, row by row as delivered: on this
        // target stores count in vmcnt like loads and return in order, so every poll round also waited for the previous round's
        // store acknowledgements -- 1.8 us per round instead of 0.7.)
        if (has_src) {
            const tw_v2 rearm{__longlong_as_double((long long)KR_TRI_SENTINEL), __longlong_as_double((long long)KR_TRI_SENTINEL)};
            for (int sidx = 7 + so; sidx < T; sidx += 4) *(__attribute__((address_space(1))) tw_v2*)(src + (int64_t)sidx * 16) = rearm;
        }
        return;
    }

    // ------------------------------------------------------------------------------------------------ the EXPORTER
    if (wave == 7) {
        // copies the rows the next workgroups need out of the solving waves' rings into the edge buffers (write-through 8-byte
        // stores), so that a solving wave's step holds no global store and no branch on where its rows go.  Four streams of 16
        // lanes: e = 0 east rows of q1 (edge lines 0-7), 1 east rows of q3 (8-15), 2 north rows of q2 (0-7), 3 north rows of q3;
        // lane idx of a stream: step offset idx >> 3 (two steps per round), edge lane idx & 7.
        const int es = l >> 4, idx = l & 15, so = idx >> 3, e = idx & 7;
        const int sq = es == 0 ? 1 : es == 2 ? 2 : 3;
        const bool east = es < 2;
        const bool on = east ? J + 1 < Q.nbj : K + 1 < Q.nbk;
        const double* const src = yring + sq * YR * 64 + (east ? 7 + 8 * e : 56 + e);
        gdouble* const dst = (gdouble*)(east ? Q.edge_e : Q.edge_n) + (int64_t)blk * edge_stride + 8 * (es & 1) + e;
        int te = on ? 0 : T;
        for (int budget = 1 << 26; budget > 0; --budget) {
            if (__ballot(te < T) == 0) break;
            const int n = max(0, min(min(tq_peek(&prog[sq]), T) - te, 2));
            if (te < T && so < n) __hip_atomic_store(dst + (int64_t)(te + so) * 16, src[((te + so) & (YR - 1)) * 64], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (te < T && n > 0) { if (idx == 0) for (int z = 0; z < n; ++z) TQ_STEP_TRACE(tq_export, es, te + z); te += n; if (idx == 0) tq_publish(&exported[es], te); }
            if (__ballot(n > 0) == 0) { if (tq_peek(&quit)) break; TQ_NAP(1); }
        }
        return;
    }

    // ------------------------------------------------------------------------------------------------ the SOLVING waves
    const int q = wave, qj = q & 1, qk = q >> 1;
    const int jl = l & 7, kl = l >> 3;
    // Where this quadrant's west / south rows come from -- a sibling's result ring (its step t + 7), a poller stream (step t), or
    // nowhere -- as DATA (pointer, step offset, ring mask, stride, counter), so that the step itself has no branch on it.
    const bool w_sib = qj == 1, w_poll = qj == 0 && J > 0, s_sib = qk == 1, s_poll = qk == 0 && K > 0;
    const double* const w_ptr = w_sib ? yring + (q - 1) * YR * 64 + 7 + 8 * kl : pring + (qk == 0 ? 0 : 1) * R * 8 + kl;
    const double* const s_ptr = s_sib ? yring + (q - 2) * YR * 64 + 56 + jl : pring + (2 + qj) * R * 8 + jl;
    const int w_off = w_sib ? 7 : 0, w_mask = w_sib ? YR - 1 : R - 1, w_stride = w_sib ? 64 : 8;
    const int s_off = s_sib ? 7 : 0, s_mask = s_sib ? YR - 1 : R - 1, s_stride = s_sib ? 64 : 8;
    int* const w_cnt = w_sib ? &prog[q - 1] : w_poll ? &pavail[qk == 0 ? 0 : 1] : &always;
    int* const s_cnt = s_sib ? &prog[q - 2] : s_poll ? &pavail[2 + qj] : &always;
    // who reads this quadrant's ring (back-pressure): sibling quadrants and exporter streams
    const int cons_a = qj == 0 ? q + 1 : -1, cons_b = qk == 0 ? q + 2 : -1;
    const int exp_a = q == 1 ? 0 : q == 3 ? 1 : -1, exp_b = q == 2 ? 2 : q == 3 ? 3 : -1;
    int* const rp_a = cons_a >= 0 ? &taken[cons_a] : &always; int* const rp_b = cons_b >= 0 ? &taken[cons_b] : &always;
    int* const rp_e = exp_a >= 0 ? &exported[exp_a] : &always; int* const rp_f = exp_b >= 0 ? &exported[exp_b] : &always;
    double* const my_ring = yring + q * YR * 64 + l;
    const int idx8 = max(l - 8, 0) * 4;
    cg_v2* const coef = (cg_v2*)Q.coef + (size_t)blk * nch * NA * 4 * TQ_LINES + 64 * q + l;

    // Coefficient requests are issued and waited for BY HAND, like the loader's: the compiler's counted vmcnt does not know how
    // many result stores lie between two requests (the two store paths differ) and then waits into the youngest request.
    struct Coef { tw_v2 a[NA][4]; };
    auto fetch = [&](Coef& cf, int kc) __attribute__((always_inline)) {
#if defined(TQ_ABL) && (TQ_ABL & 2)
        if (kc > 3) kc = kc & 3;                                           // timing only: the same four chunks again and again (cache hits)
#endif
        cg_v2* p = coef + (size_t)kc * NA * 4 * TQ_LINES;
#pragma unroll
        for (int a = 0; a < NA; ++a)
#pragma unroll
            for (int h = 0; h < 4; ++h)                                     // "+v": the request may be skipped (needs(m) below) -- same register on both paths
                asm volatile("global_load_dwordx4 %0, %1, off" : "+v"(cf.a[a][h]) : "v"(p + (a * 4 + h) * TQ_LINES) : "memory");
    };
    // Chunk m needs a request unless its coefficients equal chunk m - 3's bit for bit: those sit in its buffer already (flags from
    // setup, tri_quad_dedup_kernel; chunks past the end count as "no request").
    const uint8_t* const my_skip = skipf + q * TQ_SKIPMAX;
    // code of chunk m: bit d set = chunk m + d needs a request.  Lines longer than the table (Ni > 4 330) run without repeat flags, so
    // beyond the table the code is arithmetic (chunks past the end never need one).
    auto code_of = [&](int m) -> unsigned {
        if (m < TQ_SKIPMAX) return my_skip[m];
        return (m < nch ? 1u : 0u) | (m + 1 < nch ? 2u : 0u) | (m + 2 < nch ? 4u : 0u);
    };
    // before chunk m is computed: vector-memory operations younger than its request = the (up to) two later requests of 4 NA loads
    // each; the wave issues nothing else (its results are stored by the loader wave)
    auto arrive = [&](Coef& cf, unsigned code) __attribute__((always_inline)) {
        if (!(code & 1u)) return;                                          // nothing was requested: the values are there (and were waited for then)
        const int nf = (int)((code >> 1) & 1u) + (int)((code >> 2) & 1u);
        if (nf == 2) asm volatile("s_waitcnt vmcnt(%0)" : : "n"(8 * NA) : "memory");
        else if (nf == 1) asm volatile("s_waitcnt vmcnt(%0)" : : "n"(4 * NA) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int a = 0; a < NA; ++a)
#pragma unroll
            for (int h = 0; h < 4; ++h) asm volatile("" : "+v"(cf.a[a][h]));      // (the values exist from here on)
    };
    double y = 0.0;
    int seen = 0, stage_seen = 0, ring_safe = 0;
#ifdef KR_TW_TRACE
    long long tq_acc[7] = {0, 0, 0, 0, 0, 0, 0};
#endif
    auto process = [&](const Coef& cf, int kc) __attribute__((always_inline)) {
        const int t0 = kc * C;
        // right-hand side of the chunk off the stage
#ifdef KR_TW_TRACE
        const long long ts0 = wall_clock64();
#endif
        // (both counters only grow: what was seen last time often answers this chunk's question without another LDS round trip)
#pragma unroll 1
        for (int budget = 1 << 24; stage_seen <= kc && budget > 0; --budget) {
            stage_seen = tq_peek(&staged[q >> 1]);
            if (stage_seen <= kc) TQ_NAP(1);
        }
        TQ_ACC(12, ts0);
        const double* sp = stage + (size_t)(kc % S) * C * TQ_LINES + (64 * q + l) * 2;    // (read pair by pair: eight values held per chunk cost the backward kernel its registers)
        tw_v2 rp{0.0, 0.0};
        // back-pressure: this chunk overwrites the ring slots of steps t0 - YR .. t0 + 7 - YR: every reader must be past them
        {
#ifdef KR_TW_TRACE
            const long long tb0 = wall_clock64();
#endif
            const int need = t0 - YR + 1;                                  // a sibling reads my step p at its step p - 7, an exporter stream at p
#pragma unroll 1
            for (int budget = 1 << 24; need > ring_safe && budget > 0; --budget) {
                // five counters, ONE LDS round trip: all reads issued back to back (a reader that does not exist reads `always`)
                asm volatile("" ::: "memory");
                const int va = __hip_atomic_load(rp_a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP), vb = __hip_atomic_load(rp_b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                const int ve = __hip_atomic_load(rp_e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP), vf = __hip_atomic_load(rp_f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                const int vw = __hip_atomic_load(&written[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                asm volatile("" ::: "memory");
                int m = 8 * vw - 14;                                               // the stores of chunk w read steps >= 8 w - 7: like an exporter at that step
#if defined(TQ_ABL) && (TQ_ABL & 256)
                m = 1 << 29;                                                       // timing only: the solving wave never waits for the result stores
#endif
                if (cons_a >= 0) m = min(m, va * C);                               // (published when the chunk's steps are done)
                if (cons_b >= 0) m = min(m, vb * C);
                if (exp_a >= 0) m = min(m, ve - 7);
                if (exp_b >= 0) m = min(m, vf - 7);
                ring_safe = m;
                if (need <= ring_safe || tq_peek(&quit)) break;
                TQ_NAP(1);
            }
            TQ_ACC(13, tb0);
        }
        if (kc == 0 && q == 0 && l == 0)                                   // this block is under way: the blocks behind it may start asking
            __hip_atomic_store(&flags[blk], epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        TQ_T0(tsteps0);
        // A lone wave issues an instruction every four cycles or so, whatever its kind: the step is priced in INSTRUCTIONS (50 of them,
        // 97 with the chunk's overhead spread over its eight steps).  Addresses that the step used to derive from t -- ring slot, west /
        // south operand -- are carried instead: a chunk starts on a multiple of 8 steps and the rings hold multiples of 8, so slots
        // t0 .. t0 + 7 of the own ring, and t0 + 1 + off .. t0 + 7 + off of a source ring, never wrap inside a chunk.
        double* const ring_t0 = my_ring + (t0 & (YR - 1)) * 64;
        const double* const w_first = w_ptr + ((t0 + w_off) & w_mask) * w_stride;
        const double* const s_first = s_ptr + ((t0 + s_off) & s_mask) * s_stride;
        const double* w_next = w_ptr + ((t0 + 1 + w_off) & w_mask) * w_stride;
        const double* s_next = s_ptr + ((t0 + 1 + s_off) & s_mask) * s_stride;
#pragma unroll
        for (int u = 0; u < C; ++u) {
            const int t = t0 + u;
#if defined(TQ_ABL) && (TQ_ABL & 16)
            double yj = y, yk = y;
#else
            // the j-neighbour's row sits one lane down in the same row of 16 lanes: a DPP shift (no LDS); lanes with jl == 0 (every
            // eighth) take the west row instead.  The k-neighbour is 8 lanes down, across rows of 16: a wave permute -- requested FIRST (round 5):
            // the LDS executes a wave's operations in order and this one is on the step's dependent chain, the reads below are not
            double yj = tq_shr1(y), yk = tw_bperm(idx8, y);
#endif
#if defined(TQ_ABL) && (TQ_ABL & 8)
            const double wv = 0.5, sv = 0.25;
#else
            const double* const wp = u == 0 ? w_first : w_next;
            const double* const spn = u == 0 ? s_first : s_next;
            if (u != 0) { w_next += w_stride; s_next += s_stride; }
            asm volatile("" ::: "memory");
            double wv = *wp, sv = *spn;
            asm volatile("" ::: "memory");
            if (__builtin_expect(t >= seen, 0)) {                          // rows of the west / south neighbours for this step not yet known to be there
#ifdef KR_TW_TRACE
                const long long tw0 = wall_clock64();
#endif
                // A wave at the front of the pipeline runs in lockstep with its producers -- and its first chunks are what the next block waits
                // for.  The counters AND the rows in ONE LDS round trip (round 5): the counters are read first, so rows read behind a counter that
                // covers them are the producer's (it writes the row, then the count); read behind a counter that does not, they are read again.
#pragma unroll 1
                for (int budget = 1 << 26; budget > 0; --budget) {
                    const int cw = tq_peek(w_cnt), cs = tq_peek(s_cnt);
                    wv = *wp; sv = *spn;
                    asm volatile("" ::: "memory");
                    seen = __builtin_amdgcn_readfirstlane(min(cw - w_off, cs - s_off));
                    if (t < seen) break;
                    TQ_NAP(1);
                }
                TQ_ACC(11, tw0);
            }
#endif
            if (jl == 0) yj = wv;
            if (kl == 0) yk = sv;
            const double a1 = (u & 1) ? cf.a[0][u >> 1].y : cf.a[0][u >> 1].x, a2 = (u & 1) ? cf.a[1][u >> 1].y : cf.a[1][u >> 1].x;
            const double a3 = (u & 1) ? cf.a[2][u >> 1].y : cf.a[2][u >> 1].x;
            yj = a2 != 0.0 ? yj : 0.0;                                     // absent entry (coefficient +0.0): operand +0.0, s unchanged
            yk = a3 != 0.0 ? yk : 0.0;
            const double yi = a1 != 0.0 ? y : 0.0;
#if defined(TQ_ABL) && (TQ_ABL & 64)
            double s = 1.0;
#else
            if ((u & 1) == 0) rp = *reinterpret_cast<const tw_v2*>(sp + (u >> 1) * TQ_LINES * 2);
            double s = (u & 1) ? rp.y : rp.x;
#endif
            if (FORWARD) {                                                 // stored order: k-, j-, i-neighbour (ascending column)
                s = s - a3 * yk; s = s - a2 * yj; s = s - a1 * yi;
            } else {                                                       // i-, j-, k-neighbour, then the divisor
                s = s - a1 * yi; s = s - a2 * yj; s = s - a3 * yk;
                const double dg = (u & 1) ? cf.a[NA - 1][u >> 1].y : cf.a[NA - 1][u >> 1].x;
                s = s / dg;
            }
            y = s;
#if !(defined(TQ_ABL) && (TQ_ABL & 32))
            ring_t0[u * 64] = s;                                           // all 64 rows of the step, one unmasked store;
#endif
            if (l == 0) TQ_STEP_TRACE(tq_steps, q, t);
            tq_publish(&prog[q], t + 1);                                   // then the count (LDS executes a wave's operations in order); written by ALL lanes --
                                                                           // same value, same word: masking it down to one lane costs 15-20 ns per step (tools/micro/quadstep.hip)
        }
        TQ_ACC(16, tsteps0);
        if (l == 0) tq_publish(&taken[q], kc + 1);                       // stage slot free (same-wave LDS operations complete in order: the reads above are done)
        // (the rows stay in the ring: the loader wave of this half stores them to the caller's vector)
        if (kc < 6) TQ_STAMP(2 + kc);
        if (kc == nch - 1) TQ_STAMP(10);
    };
    // Coefficients: chunks kc + 1 and kc + 2 are in flight while chunk kc is computed (an HBM round trip is about one chunk of
    // compute once the chip is busy).  Three register buffers with STATIC roles -- the loop body is three chunks long -- because a
    // rotation by register moves makes the compiler wait for the youngest request (and for the result stores behind it) at the
    // moves: measured as one memory round trip per chunk, 2.6-2.9 us instead of ~1 (8 steps of 0.05-0.1 us plus the chunk's LDS
    // traffic).  The requests are inline asm the compiler knows nothing about (it would answer a request issued under a
    // condition with vmcnt(0) at the next use); a chunk whose coefficients repeat chunk kc - 3's is not requested at all.
    Coef c0, c1, c2;
#pragma unroll
    for (int a = 0; a < NA; ++a)
#pragma unroll
        for (int h = 0; h < 4; ++h) { c0.a[a][h] = tw_v2{0.0, 0.0}; c1.a[a][h] = tw_v2{0.0, 0.0}; c2.a[a][h] = tw_v2{0.0, 0.0}; }
#pragma unroll 1
    for (int budget = 1 << 24; tq_peek(&gate) == 0 && budget > 0; --budget) TQ_NAP(8);     // (see the loader)
    TQ_STAMP(0);
    fetch(c0, 0);                                                      // (chunks 0-2 always need their request)
    unsigned code_a = code_of(0);                                      // the code of a chunk is read one chunk ahead of its use
    if (code_a & 2u) fetch(c1, 1);
#pragma unroll 1
    for (int kc = 0; kc < nch; kc += 3) {
        TQ_T0(tf0); if (code_a & 4u) fetch(c2, kc + 2); TQ_ACC(17, tf0);
        const unsigned code_b = code_of(kc + 1);
        TQ_T0(ta0); arrive(c0, code_a); TQ_ACC(14, ta0); process(c0, kc);
        TQ_T0(tf1); if (code_b & 4u) fetch(c0, kc + 3); TQ_ACC(17, tf1);
        const unsigned code_c = code_of(kc + 2);
        if (kc + 1 < nch) { TQ_T0(ta1); arrive(c1, code_b); TQ_ACC(14, ta1); process(c1, kc + 1); }
        TQ_T0(tf2); if (code_c & 4u) fetch(c1, kc + 4); TQ_ACC(17, tf2);
        code_a = code_of(kc + 3);
        if (kc + 2 < nch) { TQ_T0(ta2); arrive(c2, code_c); TQ_ACC(14, ta2); process(c2, kc + 2); }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                   // (requests past the end: nothing of this wave's is in flight from here)
#ifdef KR_TW_TRACE
    if (FORWARD && l == 0 && blk < 4096) { for (int i = 0; i < 5; ++i) tq_trace[(blk * 4 + q) * 16 + 11 + i] = tq_acc[i]; tq_trace[(blk * 4 + q) * 16 + 1] = tq_acc[5]; tq_trace[(blk * 4 + q) * 16 + 9] = tq_acc[6]; }
#endif
    // the readers' last 7 steps want my steps T .. T + 6: rows past the end of every line, operands nobody uses
    if (l == 0) tq_publish(&prog[q], HUGE_STEPS);
}

}  // namespace kr
