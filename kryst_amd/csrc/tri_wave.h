// PIPELINED WAVEFRONT solve of a structured-grid factor (GridFactor).
//
// A workgroup owns an 8 x 8 block of grid lines (j,k).  Lane (jl,kl) of its SOLVING wave walks its line i = 0..Ni-1
// (backward: everything mirrored), one row per step, skewed by jl + kl steps, so that when it reaches row i its j- and
// k-neighbours' row i was finished one step earlier by the lanes next to it: those two values travel by wave permutes,
// the i-neighbour stays in a register.  Block-boundary lanes need the neighbour BLOCK's results, which travel through
// memory (agent-scope stores and loads; NaN sentinel = not written yet).  Blocks are dispatched in (K, J) order, so a block
// only waits for lower-numbered ones.
//
// What bounds the solve is the pipeline fill: the last block starts (Nj/8 + Nk/8 - 2) block hops after the first.  gfx950
// returns vector-memory results IN ORDER per wave, so a wave that mixes streams pays the slowest one's latency on every
// wait (coefficients from HBM ~2 us, neighbour rows ~0.6 us, write-through acknowledgements): measured with everything on
// one wave, a step cost 0.3-0.5 us and a hop 6-12 us whatever the prefetch distance.  Hence THREE waves per workgroup, tied
// together by counters in LDS (no barriers):
//   * the SOLVING wave never loads from memory.  It takes its coefficients and its neighbour rows from LDS, does the
//     arithmetic, and only issues stores, which nobody waits for: 16-byte ordinary stores once per chunk of 8 steps for
//     everybody, plus an 8-byte write-through per step for the lanes whose rows another block reads (writing every row
//     through costs ~1 us per step with 1024 waves in flight, tools/micro/steplat.hip).  Absent entries are handled by
//     selecting a +0.0 operand instead of branching (s - c*0 == s for every s when c == +0.0);
//   * the LOADER wave streams rhs / coefficients / divisor two chunks ahead through its registers into a 3-slot LDS stage;
//   * the POLLER wave reads the 16 neighbour lines (8 j-neighbours, 8 k-neighbours) 8 rows per round trip, one 16-byte load per lane, publishes the
//     leading rows that have been written into an LDS ring and asks again for the rest.  Before its producers have started
//     it only looks at their "under way" flags, rarely: a thousand waiting blocks asking for rows around the clock slow the
//     few that compute.
// The subtraction order is the stored (ascending column) order of the row: bit-identical to the level-scheduled solve.
#pragma once

namespace kr {

typedef __attribute__((address_space(1))) double gdouble;
typedef __attribute__((address_space(1))) const double cgdouble;
typedef double tw_v2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(1))) const tw_v2 cg_v2;

__device__ __forceinline__ bool tw_is_sentinel(double x) { return (unsigned long long)__double_as_longlong(x) == KR_TRI_SENTINEL; }
__device__ __forceinline__ double tw_bperm(int byte_idx, double v) {
    const int lo = __builtin_amdgcn_ds_bpermute(byte_idx, __double2loint(v));
    const int hi = __builtin_amdgcn_ds_bpermute(byte_idx, __double2hiint(v));
    return __hiloint2double(hi, lo);
}
// base (uniform) + 32-bit byte offset (per lane) + immediate
__device__ __forceinline__ cgdouble* tw_at(cgdouble* base, uint32_t off, int imm) {
    return (cgdouble*)((__attribute__((address_space(1))) const char*)base + imm + (size_t)off);
}

#ifdef KR_TW_TRACE
// trace build (tools/tw_trace.py): per block, forward solve: entry, chunk 1 done, end, poller rounds / time, gate open, first rows
__device__ long long tw_trace[8 * 4096];
__device__ long long tw_rounds[16 * 4096];     // poller: end time and rows delivered of its first 8 rounds
__device__ long long tw_steps[64 * 4096];      // per block: [0..31] time the solving wave finished step t, [32..63] time step t was published
#endif
__device__ __forceinline__ int tw_lds_load(int* p) { return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void tw_lds_store(int* p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP); }

// Sentinels only where somebody will look: the rows of the lines another block's poller reads (15 of a block's 64 lines).
// One workgroup per block of lines; also clears the "under way" flags and the abort word.
template <bool FORWARD>
__global__ __launch_bounds__(256) void tri_wave_fill_kernel(const TriArgs* args, double* out_ptr, GridView G, int32_t* flags, int32_t nflags) {
    if (args->skip) return;
    double* out = out_ptr ? out_ptr : args->z;
    const int nbj = (G.Nj + 7) >> 3, nbk = (G.Nk + 7) >> 3;
    const int J = blockIdx.x % nbj, K = blockIdx.x / nbj;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < nflags; i += gridDim.x * 256) flags[i] = 0;
    const double sentinel = __longlong_as_double((long long)KR_TRI_SENTINEL);
    for (int line = 0; line < 64; ++line) {
        const int jl = line & 7, kl = line >> 3;
        if (!((jl == 7 && J + 1 < nbj) || (kl == 7 && K + 1 < nbk))) continue;
        const int jj = J * 8 + jl, kk = K * 8 + kl;
        if (jj >= G.Nj || kk >= G.Nk) continue;
        const int j = FORWARD ? jj : G.Nj - 1 - jj, k = FORWARD ? kk : G.Nk - 1 - kk;
        double* row0 = out + (int64_t)(k * G.Nj + j) * G.Ni;
        for (int i = threadIdx.x; i < G.Ni; i += 256) row0[i] = sentinel;
    }
}

template <bool FORWARD>
__global__ __launch_bounds__(192) void tri_wave_kernel(const TriArgs* args, const double* in_ptr, double* out_ptr, GridView G, int64_t n, int32_t* flags, int32_t* abort_word,
                                                       int32_t* gave_up, int poll_budget) {
    if (args->skip) return;
    constexpr int C = 8;                                                  // steps per chunk
    constexpr int NA = FORWARD ? 4 : 5;                                   // arrays staged per chunk: rhs, c1, c2, c3 (, divisor)
    constexpr int S = 3;                                                  // LDS stage slots
    constexpr int P = 2;                                                  // chunks the loader keeps in flight in registers
    constexpr int R = 64;                                                 // steps in the neighbour ring
    __shared__ tw_v2 stage[S * NA * (C / 2) * 64];                        // [slot][array][step pair][lane]
    __shared__ double nbv[R * 16];                                        // [step % R][neighbour line]: 0-7 j-neighbours of kl, 8-15 k-neighbours of jl
    __shared__ int ctr[4];                                                // staged chunks, taken chunks, published steps, "the poller gave up"
    // every pointer in the global address space: no flat instructions
    cgdouble* in = (cgdouble*)(in_ptr ? in_ptr : args->r);
    gdouble* out = (gdouble*)(out_ptr ? out_ptr : args->z);
    cgdouble* c1 = (cgdouble*)G.c1; cgdouble* c2 = (cgdouble*)G.c2; cgdouble* c3 = (cgdouble*)G.c3; cgdouble* dgp = (cgdouble*)G.diag;
    const int wave = threadIdx.x >> 6;                                    // 0 solves, 1 loads, 2 polls
    const int l = threadIdx.x & 63;
    // the poller's lanes 4p..4p+3 stand for the boundary lane they serve: (0, p) for p < 8, (p - 8, 0) for 8 <= p < 16
    const int jl = wave == 2 ? (l < 32 ? 0 : (l >> 2) - 8) : (l & 7), kl = wave == 2 ? (l < 32 ? l >> 2 : 0) : (l >> 3), skew = jl + kl;
    const int nbj = (G.Nj + 7) >> 3, nbk = (G.Nk + 7) >> 3;
    // blocks are numbered along anti-diagonals J + K = d: the hardware starts workgroups in index order, and with a few
    // hundred of them resident at a time these must be the ones next to the front, not the first rows of the (K, J) box
    int J, K;
    {
        int d = 0, rem = blockIdx.x, lo = 0;
        for (;; ++d) {
            lo = max(0, d - (nbk - 1));
            const int cnt = min(d, nbj - 1) - lo + 1;
            if (rem < cnt) break;
            rem -= cnt;
        }
        J = lo + rem; K = d - J;
    }
    const int blk = K * nbj + J;                                          // index into the flags
    const int jj = J * 8 + jl, kk = K * 8 + kl;                           // schedule coordinates (mirrored for the backward solve)
    const bool line_ok = jj < G.Nj && kk < G.Nk;
    const bool full = J * 8 + 8 <= G.Nj && K * 8 + 8 <= G.Nk && n < ((int64_t)1 << 28);   // uniform: every lane owns a line (and 32-bit byte offsets reach every row)
    const int j = FORWARD ? jj : G.Nj - 1 - jj, k = FORWARD ? kk : G.Nk - 1 - kk;
    const int32_t s1 = G.Ni, s2 = G.Ni * G.Nj;
    const int64_t line0 = line_ok ? (int64_t)(k * G.Nj + j) * G.Ni : 0;   // row of i = 0 on this line
    const int64_t dj = FORWARD ? -(int64_t)s1 : s1, dk = FORWARD ? -(int64_t)s2 : s2;   // where the j / k neighbour's row lives
    constexpr int SG = FORWARD ? 8 : -8;                                  // bytes from one step's row to the next
    auto row_of = [&](int ii) -> int64_t { return line0 + (FORWARD ? ii : G.Ni - 1 - ii); };   // may lie outside the line (unused then)
    auto offset_of = [&](int t0) { return (uint32_t)(8 * row_of(t0 - skew)); };                 // fast chunks only (row inside the line)
    auto fast_chunk = [&](int t0) { return full && t0 >= 2 * C && t0 + C <= G.Ni; };           // every lane inside its line for all C steps
    const int nsteps = G.Ni + 14, nch = (nsteps + C - 1) / C;
    struct Chunk { double rv[C], a1[C], a2[C], a3[C], dg[C]; };           // indexed by step within the chunk
    if (threadIdx.x < 4) ctr[threadIdx.x] = 0;
    __syncthreads();                                                      // the only barrier: counters are zero before anybody looks
    int* const staged = &ctr[0]; int* const taken = &ctr[1]; int* const pub = &ctr[2];

    if (wave == 1) {
        // ---- the LOADER: chunk kc + P + 1 requested, chunk kc staged.
        // Interior chunks: FOUR lanes share a grid line's 64 contiguous bytes per array (lane 4g+c of pass r loads the 16-byte
        // piece c of line 16r+g), so one load instruction touches 16 memory lines, not 64 -- with a line per lane the loaders
        // kept each CU's address unit ~80 % busy and everybody else's requests queued behind them.  The piece goes straight
        // to the LDS row of the lane that will use it.
        struct Buf { tw_v2 d[NA][C / 2]; };                               // slow chunk: d[a][h] = this lane's steps 2h, 2h+1; fast: d[a][r] = piece of line 16r+g
        const int g = l >> 2, c = l & 3;
        uint32_t piece0[4];                                               // byte offset of (line 16r+g, step 0 of chunk 0, piece c); valid arithmetic only for fast chunks
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int L = 16 * r + g, jL = L & 7, kL = L >> 3;
            const int jx = FORWARD ? J * 8 + jL : G.Nj - 1 - (J * 8 + jL), kx = FORWARD ? K * 8 + kL : G.Nk - 1 - (K * 8 + kL);
            const int64_t row0 = (int64_t)(kx * G.Nj + jx) * G.Ni + (FORWARD ? -(jL + kL) : G.Ni - 1 + (jL + kL));   // row of step 0
            piece0[r] = (uint32_t)(8 * row0) + (FORWARD ? 0u : (uint32_t)(-8 * (C - 1))) + 16u * c;
        }
        auto fetch = [&](Buf& q, int t0) {
            if (fast_chunk(t0)) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const uint32_t off = piece0[r] + (uint32_t)(SG * t0);
                    q.d[0][r] = *(cg_v2*)tw_at(in, off, 0); q.d[1][r] = *(cg_v2*)tw_at(c1, off, 0);
                    q.d[2][r] = *(cg_v2*)tw_at(c2, off, 0); q.d[3][r] = *(cg_v2*)tw_at(c3, off, 0);
                    if (!FORWARD) q.d[NA - 1][r] = *(cg_v2*)tw_at(dgp, off, 0);
                }
            } else {                                                      // lines start / end inside the chunk, ragged blocks: element-wise, clamped
#pragma unroll
                for (int h = 0; h < C / 2; ++h) {
                    const int64_t ra = min(max(row_of(t0 + 2 * h - skew), (int64_t)0), n - 1), rb = min(max(row_of(t0 + 2 * h + 1 - skew), (int64_t)0), n - 1);
                    q.d[0][h] = tw_v2{in[ra], in[rb]}; q.d[1][h] = tw_v2{c1[ra], c1[rb]};
                    q.d[2][h] = tw_v2{c2[ra], c2[rb]}; q.d[3][h] = tw_v2{c3[ra], c3[rb]};
                    if (!FORWARD) q.d[NA - 1][h] = tw_v2{dgp[ra], dgp[rb]};
                }
            }
        };
        auto publish = [&](const Buf& q, int kc) {
            for (int budget = 1 << 24; kc - tw_lds_load(taken) >= S && budget > 0; --budget) __builtin_amdgcn_s_sleep(2);   // slot still in use
            tw_v2* dst = stage + (size_t)(kc % S) * NA * (C / 2) * 64;
            if (fast_chunk(kc * C)) {
                const int h = FORWARD ? c : C / 2 - 1 - c;                // the piece's step pair (the backward solve walks rows downwards)
#pragma unroll
                for (int a = 0; a < NA; ++a)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        dst[(a * (C / 2) + h) * 64 + 16 * r + g] = FORWARD ? q.d[a][r] : tw_v2{q.d[a][r].y, q.d[a][r].x};
            } else {
#pragma unroll
                for (int a = 0; a < NA; ++a)
#pragma unroll
                    for (int h = 0; h < C / 2; ++h) dst[(a * (C / 2) + h) * 64 + l] = q.d[a][h];
            }
            tw_lds_store(staged, kc + 1);
        };
        Buf b0, b1, b2;                                                   // chunk kc lives in buffer kc % (P + 1)
        static_assert(P == 2, "three rotating buffers");
        fetch(b0, 0);
        if (1 < nch) fetch(b1, C);
        if (2 < nch) fetch(b2, 2 * C);
        for (int kc = 0; kc < nch; kc += 3) {
            publish(b0, kc);
            if (kc + 3 < nch) fetch(b0, (kc + 3) * C);
            if (kc + 1 < nch) { publish(b1, kc + 1); if (kc + 4 < nch) fetch(b1, (kc + 4) * C); }
            if (kc + 2 < nch) { publish(b2, kc + 2); if (kc + 5 < nch) fetch(b2, (kc + 5) * C); }
        }
        return;
    }

    if (wave == 2) {
        // ---- the POLLER.  Four lanes per neighbour line, each asking for two consecutive rows with ONE 16-byte load: the
        // 8 rows of a line and round sit in one or two memory lines and travel as one or two requests.  (Uncached requests
        // to one memory line are served one after the other, ~0.11 us each: 16 single-row loads per line took 1.8 us a round.)
        const int q = l & 3;
        const bool mine = line_ok && (l < 32 ? J > 0 : K > 0);            // this lane's line has a neighbour block behind it
        const int64_t dn = l < 32 ? dj : dk;
        gdouble* const anywhere = out + __builtin_amdgcn_readfirstlane((int)(line0 >> 1)) * (int64_t)2;   // a valid pair of rows, the same for the whole wave
        // GATE: until the producers are under way, look at their flags only
        if (l == 0) {
            const bool has_w = J > 0, has_s = K > 0;
            for (int budget = 1 << 22; budget > 0; --budget) {
                const bool ok_w = !has_w || __hip_atomic_load(&flags[blk - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
                const bool ok_s = !has_s || __hip_atomic_load(&flags[blk - nbj], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
                if (ok_w && ok_s) break;
                if ((budget & 63) == 0 && __hip_atomic_load(abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) break;
                __builtin_amdgcn_s_sleep(8);
            }
        }
#ifdef KR_TW_TRACE
        if (l == 0 && FORWARD) tw_trace[8 * blk + 5] = wall_clock64();      // gate open
        bool first_pub = true;
        long long rounds = 0, empty = 0; const long long tp0 = wall_clock64();
#endif
        const int T = nch * C;
        const unsigned long long qmask = 0x1111111111111111ull;           // lanes with q == 0
        int t = 0;
        for (int budget = poll_budget; t < T && budget > 0;) {
            const int done_steps = max(tw_lds_load(taken) - 1, 0) * C;    // steps the solving wave no longer needs
            const int lim = min(T, done_steps + R);                       // ring slots free up to here
            if (t >= lim) { __builtin_amdgcn_s_sleep(2); --budget; continue; }
            // this lane: rows ii0, ii0 + 1 of its line = steps t + 2q, t + 2q + 1
            const int ii0 = t + 2 * q - skew;
            const bool n0 = mine && ii0 >= 0 && ii0 < G.Ni, n1 = mine && ii0 + 1 >= 0 && ii0 + 1 < G.Ni;
            const int w = (n0 && !n1) ? ii0 - 1 : (!n0 && n1) ? ii0 + 1 : ii0;        // a 2-row window inside the line (Ni >= 2)
            gdouble* const addr = (n0 || n1) ? &out[(FORWARD ? line0 + w : line0 + G.Ni - 2 - w) + dn] : anywhere;
            tw_v2 pair;                                                   // agent-scope (sc1) 16-byte load: each 8-byte half is one row, whole or sentinel
            asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(pair) : "v"(addr) : "memory");
            const double ra = FORWARD ? pair.x : pair.y, rb = FORWARD ? pair.y : pair.x;   // rows w, w + 1
            const double r0 = (n0 && !n1) ? rb : ra, r1 = (!n0 && n1) ? ra : rb;           // rows ii0, ii0 + 1
            const unsigned long long bad0 = __ballot(n0 && tw_is_sentinel(r0)), bad1 = __ballot(n1 && tw_is_sentinel(r1));
            int m = 0;                                                    // leading steps whose 16 rows are all there
#pragma unroll
            for (int sidx = 0; sidx < C; ++sidx) {
                const unsigned long long bad = (sidx & 1) ? bad1 : bad0;
                if (m == sidx && t + sidx < lim && (bad & (qmask << (sidx >> 1))) == 0) m = sidx + 1;
            }
            // out of patience: the rows this block waits for never came.  The pipeline only makes progress if the blocks a resident
            // block waits for are resident or finished, which holds because the hardware starts workgroups in index order -- an
            // observed property, not a HIP guarantee.  Should it ever fail (or a logic error stall the front), hand over the
            // sentinels (NaNs), tell every block to do the same so that the launch ends at once, and raise the host-visible word:
            // the host then repeats the work with the barrier-free plane kernels (ilu.hip: ilu_health)
            if (m == 0 && (budget == 1 || ((budget & 255) == 0 && __hip_atomic_load(abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0))) {
                __hip_atomic_store(abort_word, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(gave_up, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);   // mapped host word, never cleared by the device: the host reads it at its next sync
                tw_lds_store(&ctr[3], 1);
                budget = 1; m = 1;
            }
            if (2 * q < m) nbv[((t + 2 * q) % R) * 16 + (l >> 2)] = n0 ? r0 : 0.0;
            if (2 * q + 1 < m) nbv[((t + 2 * q + 1) % R) * 16 + (l >> 2)] = n1 ? r1 : 0.0;
            if (m > 0) { t += m; tw_lds_store(pub, t); }
#ifdef KR_TW_TRACE
            if (m > 0 && l == 0 && FORWARD && blk < 4096) { const long long now = wall_clock64(); for (int i = t - m; i < t && i < 32; ++i) tw_steps[64 * blk + 32 + i] = now; }
            if (m > 0 && first_pub) { first_pub = false; if (l == 0 && FORWARD) { tw_trace[8 * blk + 6] = wall_clock64(); tw_trace[8 * blk + 7] = m; } }
            if (rounds < 8 && l == 0 && FORWARD) { tw_rounds[16 * blk + 2 * rounds] = wall_clock64(); tw_rounds[16 * blk + 2 * rounds + 1] = m * 100; }
            rounds++; if (m == 0) empty++;
#endif
            if (m == 0) { __builtin_amdgcn_s_sleep(1); --budget; }
        }
#ifdef KR_TW_TRACE
        if (l == 0 && FORWARD) { tw_trace[8 * blk + 3] = rounds * 1000000 + empty; tw_trace[8 * blk + 4] = wall_clock64() - tp0; }
#endif
        return;
    }

    // ---- the SOLVING wave
#ifdef KR_TW_TRACE
    long long tr_pub = 0, tr_stage = 0;
    if (l == 0 && FORWARD) tw_trace[8 * blk + 0] = wall_clock64();
#endif
    const bool west = line_ok && jl == 0 && J > 0, south = line_ok && kl == 0 && K > 0;
    // lanes whose results another block reads: only these write through
    const bool edge = line_ok && ((jl == 7 && J + 1 < nbj) || (kl == 7 && K + 1 < nbk));
    const int idx1 = max(l - 1, 0) * 4, idx8 = max(l - 8, 0) * 4;
    const int nb_w = west ? kl : 0, nb_s = south ? 8 + jl : 0;             // this lane's entries of a ring row (anything for the others)
    double y = 0.0;
    int pub_seen = 0;
    auto take = [&](Chunk& q, int kc) {
#ifdef KR_TW_TRACE
        const long long tb0 = wall_clock64();
#endif
        for (int budget = 1 << 24; tw_lds_load(staged) <= kc && budget > 0; --budget) __builtin_amdgcn_s_sleep(1);
#ifdef KR_TW_TRACE
        tr_stage += wall_clock64() - tb0;
#endif
        const tw_v2* src = stage + (size_t)(kc % S) * NA * (C / 2) * 64 + l;
#pragma unroll
        for (int h = 0; h < C / 2; ++h) {
            const tw_v2 r2 = src[(0 * (C / 2) + h) * 64], x1 = src[(1 * (C / 2) + h) * 64], x2 = src[(2 * (C / 2) + h) * 64], x3 = src[(3 * (C / 2) + h) * 64];
            q.rv[2 * h] = r2.x; q.rv[2 * h + 1] = r2.y; q.a1[2 * h] = x1.x; q.a1[2 * h + 1] = x1.y;
            q.a2[2 * h] = x2.x; q.a2[2 * h + 1] = x2.y; q.a3[2 * h] = x3.x; q.a3[2 * h + 1] = x3.y;
            if (!FORWARD) { const tw_v2 d2v = src[(4 * (C / 2) + h) * 64]; q.dg[2 * h] = d2v.x; q.dg[2 * h + 1] = d2v.y; }
        }
        tw_lds_store(taken, kc + 1);                                      // (release: the reads above are complete)
    };
    auto neighbours = [&](int t, double& wv, double& sv) {                // rows of step t from the ring, once the poller has them
        if (t >= pub_seen) {
#ifdef KR_TW_TRACE
            const long long tb0 = wall_clock64();
#endif
            for (int budget = 1 << 26; budget > 0; --budget) {
                pub_seen = __builtin_amdgcn_readfirstlane(tw_lds_load(pub));
                if (t < pub_seen) break;
                if ((budget & 255) == 0 && __builtin_amdgcn_readfirstlane(tw_lds_load(&ctr[3])) != 0) { pub_seen = 1 << 30; break; }   // the poller gave up
                __builtin_amdgcn_s_sleep(1);
            }
#ifdef KR_TW_TRACE
            tr_pub += wall_clock64() - tb0;
#endif
        }
        const double* row = nbv + (t % R) * 16;
        wv = row[nb_w]; sv = row[nb_s];
    };
    // ---- one row
    auto row_value = [&](const Chunk& q, int u, double yj, double yk) {
        yj = q.a2[u] != 0.0 ? yj : 0.0;                                   // absent entry (coefficient +0.0): operand +0.0, s unchanged
        yk = q.a3[u] != 0.0 ? yk : 0.0;
        const double yi = q.a1[u] != 0.0 ? y : 0.0;
        double s = q.rv[u];
        if (FORWARD) {                                                    // stored order: k-, j-, i-neighbour (ascending column)
            s = s - q.a3[u] * yk; s = s - q.a2[u] * yj; s = s - q.a1[u] * yi;
        } else {                                                          // i-, j-, k-neighbour, then the divisor
            s = s - q.a1[u] * yi; s = s - q.a2[u] * yj; s = s - q.a3[u] * yk;
            s = s / q.dg[u];
        }
        return s;
    };
    // ---- 8 steps, predicate-free (every lane inside its line)
    auto run_fast = [&](const Chunk& q, int t0, uint32_t off) {
        double yv[C];
#pragma unroll
        for (int u = 0; u < C; ++u) {
            double wv, sv;
            neighbours(t0 + u, wv, sv);
            double yj = tw_bperm(idx1, y), yk = tw_bperm(idx8, y);
            if (west) yj = wv;
            if (south) yk = sv;
            y = row_value(q, u, yj, yk);
            yv[u] = y;
            if (edge) __hip_atomic_store((gdouble*)tw_at(out, off, SG * u), y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#ifdef KR_TW_TRACE
            if (l == 0 && FORWARD && t0 + u < 32 && blk < 4096) tw_steps[64 * blk + t0 + u] = wall_clock64();
#endif
        }
        if (!edge) {                                                      // (the edge lanes' rows are in memory already)
            const uint32_t lo = FORWARD ? off : off - 8 * (C - 1);       // lowest row of the chunk
#pragma unroll
            for (int h = 0; h < C / 2; ++h) {
                tw_v2 v; v.x = yv[FORWARD ? 2 * h : C - 1 - 2 * h]; v.y = yv[FORWARD ? 2 * h + 1 : C - 2 - 2 * h];
                *(__attribute__((address_space(1))) tw_v2*)tw_at(out, lo, 16 * h) = v;
            }
        }
    };
    // ---- 8 steps, every predicate (lines that start / end inside the chunk, ragged blocks)
    auto run_any = [&](const Chunk& q, int t0) {
#pragma unroll
        for (int u = 0; u < C; ++u) {
            const int ii = t0 + u - skew;
            const bool act = line_ok && ii >= 0 && ii < G.Ni;
            const int64_t row = row_of(ii);
            double wv, sv;
            neighbours(t0 + u, wv, sv);
            double yj = tw_bperm(idx1, y), yk = tw_bperm(idx8, y);
            if (west) yj = wv;
            if (south) yk = sv;
            const double s = row_value(q, u, yj, yk);
            if (act) {
                y = s;
                if (edge) __hip_atomic_store(&out[row], s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                else out[row] = s;
            }
            if (t0 == 0 && u == 0 && l == 0)                              // this block is under way: the blocks behind it may start asking
                __hip_atomic_store(&flags[blk], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#ifdef KR_TW_TRACE
            if (l == 0 && FORWARD && t0 + u < 32 && blk < 4096) tw_steps[64 * blk + t0 + u] = wall_clock64();
#endif
        }
    };
    Chunk q;
    for (int kc = 0; kc < nch; ++kc) {
        take(q, kc);
        if (fast_chunk(kc * C)) run_fast(q, kc * C, offset_of(kc * C)); else run_any(q, kc * C);
#ifdef KR_TW_TRACE
        if (l == 0 && FORWARD && kc == 1) tw_trace[8 * blk + 1] = wall_clock64();
#endif
    }
#ifdef KR_TW_TRACE
    if (l == 0 && FORWARD) { tw_trace[8 * blk + 2] = wall_clock64(); }
#endif
}

}  // namespace kr
