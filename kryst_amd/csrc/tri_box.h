// PIPELINED WAVEFRONT solve of a box-stencil factor (BoxFactor: up to 13 entries per row inside the 3 x 3 x 3 cube) -- tri_wave.h's
// line pipeline with the skew and the neighbour rings a 27-point stencil needs (round 4).
//
// In SCHEDULE coordinates (the backward solve mirrors i, j and k) row (ii, jj, kk) needs rows ii-1 .. ii+1 of the lines (jj-1, kk),
// (jj+1, kk-1), (jj, kk-1), (jj-1, kk-1) and row ii-1 of its own line: everything on hyperplanes ii + 2 jj + 4 kk that came earlier.  A
// workgroup owns a PARALLELOGRAM of 8 x 8 lines: lane (jl, kl) of its solving wave walks line jj = 8 J + jl - kl, kk = 8 K + kl, one row per
// step, skewed by 2 (jl + kl) + 1 steps.  The shear turns the four neighbour lines into the lanes (jl-1, kl), (jl, kl-1), (jl-1, kl-1),
// (jl-2, kl-1): never a lane to the right, so a block only needs blocks (J-1, K), (J, K-1), (J+1, K-1) -- all earlier in the dispatch
// order J + 2 K -- and inside a block every value travels by wave permutes: at step t a lane receives row ii+1 of each neighbour line
// (the neighbour finished it 1, 1, 3 and 5 steps ago: it hands over its value of that age) and keeps the two rows before it in
// registers.  25 lines of other blocks feed the lanes on the parallelogram's left and lower sides: a poller wave reads them (NaN
// sentinel = not written yet) into an LDS ring indexed by ROW, since up to three lanes with different skews read the same line.  Two
// loader waves stream the right-hand side (from the vector) and the 13 coefficient streams and the divisor (from a BLOCKED copy laid out
// exactly like the LDS stage: tri_box_layout_kernel) two chunks ahead into a two-slot LDS stage (the solving wave copies a chunk into
// registers at once, which frees the slot).  No barriers after the first; every wait has a budget.  Specialisations: REGULAR (no
// absent-entry selects), ALL (the factor has all 13 streams: no per-stream tests); a stream a factor does not have is neither staged nor
// subtracted.
//
// Subtraction order: the stored (ascending column) order of the row -- forward: schedule offsets (-1,-1,-1) ... (0,0,-1) = streams 0..12;
// backward: stream a is the schedule offset with code 12 - a.  Absent entries select a +0.0 operand (tri_wave.h).  Same bits as the
// level-scheduled solve and as tri_box_plane_kernel, the barrier-free form this one falls back to when a poller runs out of patience.
#pragma once

namespace kr {

constexpr int TB_C = 4;                 // steps per chunk
constexpr int TB_R = 64;                // rows in the neighbour ring
constexpr int TB_RS = 26;               // ring row stride: 25 external lines
constexpr int TB_S = 2;                 // LDS stage slots
template <bool FORWARD> constexpr int tb_arrays() { return FORWARD ? 14 : 15; }              // rhs, 13 streams (, divisor)
template <bool FORWARD> constexpr size_t tb_lds_bytes() { return (size_t)TB_S * tb_arrays<FORWARD>() * (TB_C / 2) * 64 * 16 + (size_t)TB_R * TB_RS * 8 + 64; }

__host__ __device__ __forceinline__ int tb_nbj(int Nj) { return (Nj + 6) / 8 + 1; }         // J = 0 .. : line jj = 8 J + jl - kl covers -7 .. Nj - 1
__host__ __device__ __forceinline__ int tb_nbk(int Nk) { return (Nk + 7) / 8; }

// blocks are numbered along d = J + 2 K (K ascending inside a d): the hardware starts workgroups in index order
__device__ __forceinline__ void tb_block_of(int b, int nbj, int nbk, int& J, int& K) {
    int rem = b;
    for (int d = 0; d < nbj + 2 * nbk; ++d) {
        const int lo = max(0, (d - (nbj - 1) + 1) >> 1), hi = min(nbk - 1, d >> 1);
        const int cnt = hi - lo + 1;
        if (cnt <= 0) continue;
        if (rem < cnt) { K = lo + rem; J = d - 2 * K; return; }
        rem -= cnt;
    }
    J = 0; K = 0;                                                          // (not reached: b < nbj nbk)
}

// Sentinels only where somebody will look: the lines another block's poller reads (lanes jl >= 6 and kl == 7); clears the "under way" flags.
template <bool FORWARD>
__global__ __launch_bounds__(256) void tri_box_fill_kernel(const TriArgs* args, double* out_ptr, BoxView B, int32_t* flags, int32_t nflags) {
    if (args->skip) return;
    double* out = out_ptr ? out_ptr : args->z;
    const int nbj = tb_nbj(B.Nj), nbk = tb_nbk(B.Nk);
    const int J = blockIdx.x % nbj, K = blockIdx.x / nbj;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < nflags; i += gridDim.x * 256) flags[i] = 0;
    const double sentinel = __longlong_as_double((long long)KR_TRI_SENTINEL);
    for (int line = 0; line < 64; ++line) {
        const int jl = line & 7, kl = line >> 3;
        if (!((jl >= 6 && J + 1 < nbj) || (kl == 7 && K + 1 < nbk))) continue;
        const int jj = J * 8 + jl - kl, kk = K * 8 + kl;
        if (jj < 0 || jj >= B.Nj || kk >= B.Nk) continue;
        const int j = FORWARD ? jj : B.Nj - 1 - jj, k = FORWARD ? kk : B.Nk - 1 - kk;
        double* row0 = out + ((int64_t)k * B.Nj + j) * B.Ni;
        for (int i = threadIdx.x; i < B.Ni; i += 256) row0[i] = sentinel;
    }
}

// BLOCKED COPY of a factor's coefficients (and divisor): [block][chunk][stream][step pair][lane] sixteen-byte elements -- exactly what a loader
// wave puts into the LDS stage, in that order, zeros (divisor: 1) where a lane is outside its line.  From the natural-order streams a loader's
// 64 lanes fetch 32 pieces of 32 bytes from 32 different memory lines per instruction, and whether the other 96 bytes of each line are
// still in L2 when the next three chunks want them depends on how many blocks are active: the loads were 29 % of the 96^3 apply and 49 % of the
// 128^3 one (profiles/r04/box_step_ablations.txt).  From the blocked copy an instruction reads 1 KB of consecutive bytes, each byte once.
template <bool FORWARD>
__global__ __launch_bounds__(256) void tri_box_layout_kernel(BoxView B, tw_v2* cb) {
    constexpr int C = TB_C, NAc = FORWARD ? 13 : 14;
    const int nbj = tb_nbj(B.Nj);
    const int nch = (B.Ni + 29 + C - 1) / C;
    const int blk = blockIdx.x / nch, kc = blockIdx.x % nch, J = blk % nbj, K = blk / nbj;
    for (int idx = threadIdx.x; idx < NAc * (C / 2) * 64; idx += 256) {
        const int lane = idx & 63, h = (idx >> 6) % (C / 2), a = idx / (64 * (C / 2));
        const int jl = lane & 7, kl = lane >> 3, jj = 8 * J + jl - kl, kk = 8 * K + kl, skew = 2 * (jl + kl) + 1;
        const bool ok = jj >= 0 && jj < B.Nj && kk < B.Nk;
        const int j = FORWARD ? jj : B.Nj - 1 - jj, k = FORWARD ? kk : B.Nk - 1 - kk;
        const int64_t base = ok ? ((int64_t)k * B.Nj + j) * B.Ni + (FORWARD ? 0 : B.Ni - 1) : 0;
        double v[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int ii = kc * C + 2 * h + e - skew;
            const bool valid = ok && ii >= 0 && ii < B.Ni;
            const int64_t row = base + (FORWARD ? ii : -ii);
            v[e] = valid ? (a < 13 ? B.c[(int64_t)a * B.cs + row] : B.diag[row]) : (a < 13 ? 0.0 : 1.0);
        }
        cb[(size_t)blockIdx.x * (NAc * (C / 2) * 64) + idx] = tw_v2{v[0], v[1]};
    }
}

// REGULAR: every stream of the factor is either absent altogether (`present` bit clear: the term is skipped) or has an entry wherever the
// neighbour row exists in the box -- the factors of a full 27- or 19-point stencil, Ilup(1) of a 7-point operator, ...  Then an absent entry's
// operand is always a row OUTSIDE the box, and those are +0.0 by construction here (the poller hands +0.0 for rows outside a line, a lane
// outside its line publishes +0.0): +0.0 x +0.0 = +0.0 leaves s unchanged bit for bit, so the 13 compare + select pairs of the general form
// are not needed.  A factor with entries missing in the interior (dropped couplings, values that cancelled to zero) takes REGULAR = false.
// ALL: the factor has all 13 streams (`present` == 0x1fff): no per-stream tests anywhere ("skip this term" costs two selects per term ON the
// dependent chain of s, present or not: 0.39 -> 0.32 us per step without them).
// MASK: the factor's streams when known at compile time -- 0x1fff (ALL), a 19-point stencil's (lower 0x1eba, upper 0x0baf), the two patterns
// Ilup(1) makes of a 7-point operator (0x1cb0, 0x01a7) -- or 0: the launch's `present` decides at run time (selects).
template <bool FORWARD, bool REGULAR, uint32_t MASK>
__global__ __launch_bounds__(256) void tri_box_kernel(const TriArgs* args, const double* in_ptr, double* out_ptr, BoxView B, int32_t* flags, int32_t* abort_word,
                                                      int32_t* gave_up, int poll_budget, uint32_t present) {
    if (args->skip) return;
    constexpr int C = TB_C, NA = tb_arrays<FORWARD>(), S = TB_S, R = TB_R, RS = TB_RS;
    constexpr int LPL = C / 2;                                            // loader, fast path: lanes per line and chunk (16-byte pieces) = passes
    constexpr int LPP = 64 / LPL;                                         // lines per pass
    constexpr int NA1 = 8;                                                // arrays of the first loader wave (the second takes the rest)
    extern __shared__ __attribute__((aligned(16))) char tb_smem[];
    tw_v2* const stage = (tw_v2*)tb_smem;                                 // [slot][array][step pair][lane]
    double* const ring = (double*)(tb_smem + (size_t)S * NA * (C / 2) * 64 * 16);   // [row % R][external line]
    int* const ctr = (int*)(ring + R * RS);                               // staged (loader 1), staged (loader 2), taken, published, "the poller gave up"
    cgdouble* in = (cgdouble*)(in_ptr ? in_ptr : args->r);
    gdouble* out = (gdouble*)(out_ptr ? out_ptr : args->z);
    const int64_t n = B.n;
    constexpr bool ALL = MASK == 0x1fffu;
    if (MASK != 0) present = MASK;                                        // (compile-time streams: every test on `present` folds away)
    const int wave = threadIdx.x >> 6;                                    // 0 solves, 1 and 2 load, 3 polls
    const int l = threadIdx.x & 63;
    const int nbj = tb_nbj(B.Nj), nbk = tb_nbk(B.Nk);
    int J, K;
    tb_block_of(blockIdx.x, nbj, nbk, J, K);
    const int blk = K * nbj + J;                                          // index into the flags
    // memory row of schedule row ii = 0 of schedule line (jj, kk); schedule row ii lives ii rows further up (forward) / down (backward)
    auto line_base = [&](int jj, int kk) -> int64_t {
        const int j = FORWARD ? jj : B.Nj - 1 - jj, k = FORWARD ? kk : B.Nk - 1 - kk;
        return ((int64_t)k * B.Nj + j) * B.Ni + (FORWARD ? 0 : B.Ni - 1);
    };
    auto line_exists = [&](int jj, int kk) { return jj >= 0 && jj < B.Nj && kk >= 0 && kk < B.Nk; };
    const int nsteps = B.Ni + 29, nch = (nsteps + C - 1) / C, T = nch * C;
    if (threadIdx.x < 8) ctr[threadIdx.x] = 0;
    // The ring starts as +0.0: a lane with skew s asks for rows -s+1 .. -1 of its external lines during its first steps, rows the poller never
    // delivers (it starts at row 0).  Their coefficients are +0.0, but a REGULAR factor multiplies without a select, and whatever the LDS
    // held before -- a NaN, on one GPU box in five -- would end up in s.
    for (int x = threadIdx.x; x < R * RS; x += blockDim.x) ring[x] = 0.0;
    __syncthreads();                                                      // the only barrier: counters and ring are zero before anybody looks
    int* const staged1 = &ctr[0]; int* const staged2 = &ctr[1]; int* const taken = &ctr[2]; int* const pub = &ctr[3]; int* const quit = &ctr[4];

    if (wave == 1 || wave == 2) {
        // ---- a LOADER: chunk kc + 3 requested, chunk kc staged.  Fast path: LPL lanes share a line's C * 8 contiguous bytes per array
        // (lane LPL g + c of pass r loads the 16-byte piece c of line LPP r + g); rows outside a line (it starts / ends inside the chunk)
        // are rows of the lines next to it in memory: loaded, staged and never used.  Only a piece outside the ARRAY (first / last
        // lines of the box) sends the whole chunk down the element-wise path.
        const int a0 = wave == 1 ? 0 : NA1, cnt = wave == 1 ? NA1 : NA - NA1;
        // arrays this wave moves: the right-hand side, the divisor, and the streams the factor HAS (what bounds a step is the bytes one CU can
        // pull per microsecond -- 15 arrays x 64 lines x 8 bytes per step: 0.38 us at ~20 GB/s -- so a stream without entries is not streamed)
        auto wanted = [&](int a) { const int g = a0 + a; return a < cnt && (ALL || g == 0 || g == 14 || ((present >> (g - 1)) & 1u)); };
        int* const staged = wave == 1 ? staged1 : staged2;
        struct Buf { tw_v2 d[NA1][C / 2]; };
        const int g = l / LPL, c = l % LPL;
        int64_t row0[LPL]; bool lok[LPL];                                 // (line LPP r + g): memory row of step 0, exists
#pragma unroll
        for (int r = 0; r < LPL; ++r) {
            const int L = LPP * r + g, jL = L & 7, kL = L >> 3, jj = 8 * J + jL - kL, kk = 8 * K + kL, sk = 2 * (jL + kL) + 1;
            lok[r] = line_exists(jj, kk);
            row0[r] = lok[r] ? line_base(jj, kk) + (FORWARD ? -sk : sk) : 0;
        }
        // own line, for the element-wise path
        const int jl = l & 7, kl = l >> 3, skew = 2 * (jl + kl) + 1;
        const bool line_ok = line_exists(8 * J + jl - kl, 8 * K + kl);
        const int64_t base = line_ok ? line_base(8 * J + jl - kl, 8 * K + kl) : 0;
        auto is_fast = [&](int t0) -> bool {
            bool ok = n < ((int64_t)1 << 28);
#pragma unroll
            for (int r = 0; r < LPL; ++r) {
                const int64_t lo = row0[r] + (FORWARD ? t0 + 2 * c : -(t0 + C - 1) + 2 * c);      // lower row of this lane's piece
                ok = ok && (!lok[r] || (lo >= 0 && lo + 1 < n));
            }
            return __all(ok);
        };
        // coefficients and divisor: the blocked copy (tri_box_layout_kernel), element (stream, step pair, lane) of this block's chunk
        constexpr int NAc = FORWARD ? 13 : 14;
        cg_v2* const cb_blk = (cg_v2*)B.cb + (size_t)blk * nch * (NAc * (C / 2) * 64) + l;
        auto fetch = [&](Buf& q, int t0) {
            cg_v2* const cbk = cb_blk + (size_t)(t0 / C) * (NAc * (C / 2) * 64);
#pragma unroll
            for (int a = 0; a < NA1; ++a)
                if (a0 + a >= 1 && wanted(a))
#pragma unroll
                    for (int h = 0; h < C / 2; ++h) q.d[a][h] = cbk[((a0 + a - 1) * (C / 2) + h) * 64];
            if (a0 != 0) return;
            // the right-hand side (array 0) changes with every apply: from the vector itself, as before
            if (is_fast(t0)) {
#pragma unroll
                for (int r = 0; r < LPL; ++r) {
                    const int64_t lo = lok[r] ? row0[r] + (FORWARD ? t0 + 2 * c : -(t0 + C - 1) + 2 * c) : 0;
                    q.d[0][r] = *(cg_v2*)tw_at(in, (uint32_t)(8 * lo), 0);
                }
            } else {
#pragma unroll
                for (int h = 0; h < C / 2; ++h) {
                    const int64_t ra = min(max(base + (FORWARD ? 1 : -1) * (int64_t)(t0 + 2 * h - skew), (int64_t)0), n - 1);
                    const int64_t rb = min(max(base + (FORWARD ? 1 : -1) * (int64_t)(t0 + 2 * h + 1 - skew), (int64_t)0), n - 1);
                    q.d[0][h] = tw_v2{in[ra], in[rb]};
                }
            }
        };
        auto publish = [&](const Buf& q, int kc) {
            for (int budget = 1 << 24; kc - tw_lds_load(taken) >= S && budget > 0; --budget) __builtin_amdgcn_s_sleep(2);   // slot still in use
            tw_v2* dst = stage + (size_t)(kc % S) * NA * (C / 2) * 64;
#pragma unroll
            for (int a = 0; a < NA1; ++a)
                if (a0 + a >= 1 && wanted(a))
#pragma unroll
                    for (int h = 0; h < C / 2; ++h) dst[((a0 + a) * (C / 2) + h) * 64 + l] = q.d[a][h];
            if (a0 == 0) {
                if (is_fast(kc * C)) {
                    const int h = FORWARD ? c : C / 2 - 1 - c;            // the piece's step pair (the backward solve walks rows downwards)
#pragma unroll
                    for (int r = 0; r < LPL; ++r) dst[h * 64 + LPP * r + g] = FORWARD ? q.d[0][r] : tw_v2{q.d[0][r].y, q.d[0][r].x};
                } else {
#pragma unroll
                    for (int h = 0; h < C / 2; ++h) dst[h * 64 + l] = q.d[0][h];
                }
            }
            tw_lds_store(staged, kc + 1);
        };
        Buf b0, b1, b2;                                                   // chunk kc lives in buffer kc % 3
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

    if (wave == 3) {
        // ---- the POLLER.  External line e: 0-7 the lines of lanes (7, e) of block (J-1, K); 8-14 those of its lanes (6, e-8); 15-24 the
        // lines jj = 8 J - 1 + (e - 15) of the row kk = 8 K - 1 (lanes (6,7), (7,7) of block (J, K-1) and (0..7, 7) of block (J+1, K-1)).
        // A row r of line e is first needed at step r + sig(e).  Four lanes per line, each asking for two consecutive rows with ONE
        // 16-byte load; slot A of a lane serves line l / 4, slot B line 16 + l / 4 (l < 36).
        const int q = l & 3;
        int64_t ebase[2]; bool eon[2]; int esig[2], eidx[2];
#pragma unroll
        for (int sl = 0; sl < 2; ++sl) {
            const int e = sl * 16 + (l >> 2);
            int jj, kk, sig;
            if (e < 8) { jj = 8 * J - 1 - e; kk = 8 * K + e; sig = 2 * e; }
            else if (e < 15) { jj = 8 * J - 2 - (e - 8); kk = 8 * K + (e - 8); sig = 2 * (e - 8) + 2; }
            else { const int m = e - 15; jj = 8 * J - 1 + m; kk = 8 * K - 1; sig = m <= 1 ? 0 : 2 * m - 4; }
            eon[sl] = e < 25 && line_exists(jj, kk);
            ebase[sl] = eon[sl] ? line_base(jj, kk) : 0;
            esig[sl] = sig; eidx[sl] = min(e, 24);
        }
        gdouble* const anywhere = out;                                    // a valid pair of rows (n >= 27)
        // GATE: until the producers are under way, look at their flags only
        if (l == 0) {
            const bool has_w = J > 0, has_s = K > 0, has_se = K > 0 && J + 1 < nbj;
            for (int budget = 1 << 22; budget > 0; --budget) {
                const bool ok_w = !has_w || __hip_atomic_load(&flags[blk - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
                const bool ok_s = !has_s || __hip_atomic_load(&flags[blk - nbj], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
                const bool ok_se = !has_se || __hip_atomic_load(&flags[blk - nbj + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
                if (ok_w && ok_s && ok_se) break;
                if ((budget & 63) == 0 && __hip_atomic_load(abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) break;
                __builtin_amdgcn_s_sleep(8);
            }
        }
        const unsigned long long qmask = 0x1111111111111111ull;           // lanes with q == 0
        int t = 0;
        for (int budget = poll_budget; t < T && budget > 0;) {
            const int done_steps = max(tw_lds_load(taken) - 1, 0) * C;    // steps the solving wave no longer needs
            const int lim = min(T, done_steps + R - 8);                   // ring rows free up to here (a row is read until 5 steps after its first use)
            if (t >= lim) { __builtin_amdgcn_s_sleep(2); --budget; continue; }
            double r0[2], r1[2]; bool n0[2], n1[2];
            tw_v2 pair[2];
            gdouble* addr[2];
            bool flip[2][2];
#pragma unroll
            for (int sl = 0; sl < 2; ++sl) {
                const int ii0 = t - esig[sl] + 2 * q;                     // this lane: rows ii0, ii0 + 1 of its line = steps t + 2q, t + 2q + 1
                n0[sl] = eon[sl] && ii0 >= 0 && ii0 < B.Ni; n1[sl] = eon[sl] && ii0 + 1 >= 0 && ii0 + 1 < B.Ni;
                const int w = (n0[sl] && !n1[sl]) ? ii0 - 1 : (!n0[sl] && n1[sl]) ? ii0 + 1 : ii0;   // a 2-row window inside the line (Ni >= 2)
                addr[sl] = (n0[sl] || n1[sl]) ? &out[FORWARD ? ebase[sl] + w : ebase[sl] - w - 1] : anywhere;
                flip[sl][0] = n0[sl] && !n1[sl]; flip[sl][1] = !n0[sl] && n1[sl];
            }
            // agent-scope (sc1) 16-byte loads: each 8-byte half is one row, whole or sentinel
            asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(pair[0]) : "v"(addr[0]) : "memory");
            asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(pair[1]) : "v"(addr[1]) : "memory");
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(pair[0]), "+v"(pair[1]) : : "memory");
            unsigned long long bad0 = 0, bad1 = 0;
#pragma unroll
            for (int sl = 0; sl < 2; ++sl) {
                const double ra = FORWARD ? pair[sl].x : pair[sl].y, rb = FORWARD ? pair[sl].y : pair[sl].x;   // rows w, w + 1
                r0[sl] = flip[sl][0] ? rb : ra; r1[sl] = flip[sl][1] ? ra : rb;                               // rows ii0, ii0 + 1
                bad0 |= __ballot(n0[sl] && tw_is_sentinel(r0[sl])); bad1 |= __ballot(n1[sl] && tw_is_sentinel(r1[sl]));
            }
            int m = 0;                                                    // leading steps whose rows are all there
#pragma unroll
            for (int sidx = 0; sidx < 8; ++sidx) {
                const unsigned long long bad = (sidx & 1) ? bad1 : bad0;
                if (m == sidx && t + sidx < lim && (bad & (qmask << (sidx >> 1))) == 0) m = sidx + 1;
            }
            // out of patience (tri_wave.h): hand over whatever is there, tell every block to do the same, raise the host-visible word; the
            // host repeats the work with the barrier-free plane kernels (ilu.hip: ilu_health)
            if (m == 0 && (budget == 1 || ((budget & 255) == 0 && __hip_atomic_load(abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0))) {
                __hip_atomic_store(abort_word, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(gave_up, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                tw_lds_store(quit, 1);
                budget = 1; m = 1;
            }
#pragma unroll
            for (int sl = 0; sl < 2; ++sl) {
                if (sl == 1 && l >= 36) break;
                const int row = t - esig[sl] + 2 * q;
                if (2 * q < m) ring[((row) & (R - 1)) * RS + eidx[sl]] = n0[sl] ? r0[sl] : 0.0;
                if (2 * q + 1 < m) ring[((row + 1) & (R - 1)) * RS + eidx[sl]] = n1[sl] ? r1[sl] : 0.0;
            }
            if (m > 0) { t += m; tw_lds_store(pub, t); }
            else { __builtin_amdgcn_s_sleep(1); --budget; }
        }
        return;
    }

    // ---- the SOLVING wave
    const int jl = l & 7, kl = l >> 3, skew = 2 * (jl + kl) + 1;
    const int jj = 8 * J + jl - kl, kk = 8 * K + kl;
    const bool line_ok = line_exists(jj, kk);
    const int64_t base = line_ok ? line_base(jj, kk) : 0;
    const bool edge = line_ok && ((jl >= 6 && J + 1 < nbj) || (kl == 7 && K + 1 < nbk));      // lanes whose rows another block reads: these write through
    const int i1 = max(l - 1, 0) * 4, i2 = max(l - 8, 0) * 4, i3 = max(l - 9, 0) * 4, i4 = max(l - 10, 0) * 4;
    // where a neighbour line is another block's: its column of the ring (-1: a lane of this wave)
    const int e1 = jl == 0 ? kl : -1;
    const int e2 = kl == 0 ? 17 + jl : -1;
    const int e3 = kl == 0 ? 16 + jl : (jl == 0 ? kl - 1 : -1);
    const int e4 = kl == 0 ? 15 + jl : (jl == 0 ? 7 + kl : (jl == 1 ? kl - 1 : -1));
    const int c1 = max(e1, 0), c2 = max(e2, 0), c3 = max(e3, 0), c4 = max(e4, 0);
    double yh0 = 0.0, yh1 = 0.0, yh2 = 0.0, yh3 = 0.0, yh4 = 0.0;        // this lane's results of the last five steps (yh0: the last one)
    double p1a = 0.0, p1b = 0.0, p2a = 0.0, p2b = 0.0, p3a = 0.0, p3b = 0.0, p4a = 0.0, p4b = 0.0;   // rows ii, ii - 1 of the four neighbour lines
    int pub_seen = 0;
    struct Chunk { double v[NA][C]; };
    auto take = [&](Chunk& q, int kc) {
        for (int budget = 1 << 24; (tw_lds_load(staged1) <= kc || tw_lds_load(staged2) <= kc) && budget > 0; --budget) __builtin_amdgcn_s_sleep(1);
        const tw_v2* src = stage + (size_t)(kc % S) * NA * (C / 2) * 64 + l;
#pragma unroll
        for (int a = 0; a < NA; ++a) {
            if (!ALL && a >= 1 && a <= 13 && !((present >> (a - 1)) & 1u)) continue;   // (not staged: the factor has no such stream)
#pragma unroll
            for (int h = 0; h < C / 2; ++h) { const tw_v2 x = src[(a * (C / 2) + h) * 64]; q.v[a][2 * h] = x.x; q.v[a][2 * h + 1] = x.y; }
        }
        tw_lds_store(taken, kc + 1);                                      // (release: the reads above are complete)
    };
    Chunk q;
    for (int kc = 0; kc < nch; ++kc) {
        take(q, kc);
#pragma unroll
        for (int u = 0; u < C; ++u) {
            const int t = kc * C + u, ii = t - skew;
            const bool act = line_ok && ii >= 0 && ii < B.Ni;
            if (t >= pub_seen) {                                          // the poller has this step's rows
                for (int budget = 1 << 26; budget > 0; --budget) {
                    pub_seen = __builtin_amdgcn_readfirstlane(tw_lds_load(pub));
                    if (t < pub_seen) break;
                    if ((budget & 255) == 0 && __builtin_amdgcn_readfirstlane(tw_lds_load(quit)) != 0) { pub_seen = 1 << 30; break; }
                    __builtin_amdgcn_s_sleep(1);
                }
            }
            // row ii + 1 of the four neighbour lines: asked for here, looked at after the products that do not need them
            double r1 = tw_bperm(i1, yh0), r2 = tw_bperm(i2, yh0), r3 = tw_bperm(i3, yh2), r4 = tw_bperm(i4, yh4);
            const double* rr = ring + ((ii + 1) & (R - 1)) * RS;
            const double x1 = rr[c1], x2 = rr[c2], x3 = rr[c3], x4 = rr[c4];
            // one term of the row: coefficient of stream a times its operand (general form, absent entry -- coefficient +0.0: the operand's
            // high word cleared -> zero or a positive subnormal, the product is +0.0 and s unchanged, whatever the operand was)
            auto product = [&](int a, double x) -> double {
                const double cf = q.v[1 + a][u];
                if (!REGULAR) x = __hiloint2double(cf != 0.0 ? __double2hiint(x) : 0, __double2loint(x));
                return cf * x;
            };
            // operands by schedule offset code 9 (dk + 1) + 3 (dj + 1) + (di + 1); codes 2, 5, 8, 11 arrive this step
            const double old_dep[13] = {p4b, p4a, 0.0, p3b, p3a, 0.0, p2b, p2a, 0.0, p1b, p1a, 0.0, yh0};
            double pr[13];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int a = 0; a < 13; ++a) {                                // the nine products of operands this lane already holds
                const int code = FORWARD ? a : 12 - a;
                if (code != 2 && code != 5 && code != 8 && code != 11) pr[a] = product(a, old_dep[code]);
            }
            __builtin_amdgcn_sched_barrier(0);
            r1 = e1 >= 0 ? x1 : r1; r2 = e2 >= 0 ? x2 : r2; r3 = e3 >= 0 ? x3 : r3; r4 = e4 >= 0 ? x4 : r4;
#pragma unroll
            for (int a = 0; a < 13; ++a) {                                // the four that wait for the arrivals
                const int code = FORWARD ? a : 12 - a;
                if (code == 2) pr[a] = product(a, r4);
                if (code == 5) pr[a] = product(a, r3);
                if (code == 8) pr[a] = product(a, r2);
                if (code == 11) pr[a] = product(a, r1);
            }
            double s = q.v[0][u];
#pragma unroll
            for (int a = 0; a < 13; ++a) {                                // subtracted in the stored order
                // (uniform: a stream the factor does not have is not staged either.  The compiler turns the skip into two selects per term; forced
                // to be a branch it was slower -- Ilup(1) of the 7-point operator on 128^3, 6 of 13 streams: 0.96 ms with selects, 1.12 with branches)
                if (!ALL && !((present >> a) & 1u)) continue;
                s = s - pr[a];
            }
            if (!FORWARD) s = s / q.v[NA - 1][u];
            if (REGULAR) s = act ? s : 0.0;                               // a row outside the box: +0.0 for whoever reads it
            if (act) {
                gdouble* dst = out + (base + (FORWARD ? ii : -ii));
                if (edge) __hip_atomic_store(dst, s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                else *dst = s;
            }
            if (t == 0 && l == 0)                                         // this block is under way: the blocks behind it may start asking
                __hip_atomic_store(&flags[blk], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            yh4 = yh3; yh3 = yh2; yh2 = yh1; yh1 = yh0; yh0 = s;
            p1b = p1a; p1a = r1; p2b = p2a; p2a = r2; p3b = p3a; p3a = r3; p4b = p4a; p4a = r4;
        }
    }
}

}  // namespace kr
