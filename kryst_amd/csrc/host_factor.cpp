// Host-side factorisations of the ILU family (host_factor.h): no device call, no HIP header -- this file also builds for the CPU alone
// under ThreadSanitizer / AddressSanitizer (`make san`).
#include "host_factor.h"
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <pthread.h>
#include <sched.h>

namespace kr {

// ---- pool of big host blocks (KRYST_HOST_POOL_MB, default 2048; 0: off)
namespace {
struct HostPool {
    std::mutex mu; std::vector<std::pair<void*, size_t>> blocks; size_t held = 0;
    ~HostPool() { for (auto& b : blocks) std::free(b.first); }
};
HostPool g_host_pool;
size_t host_pool_limit() {
    const char* e = getenv("KRYST_HOST_POOL_MB");
    const long long mb = e ? atoll(e) : 2048;
    return mb <= 0 ? 0 : (size_t)mb << 20;
}
}  // namespace
void* host_pool_take(size_t bytes) {
    HostPool& P = g_host_pool;
    std::lock_guard<std::mutex> g(P.mu);
    size_t best = P.blocks.size();
    for (size_t i = 0; i < P.blocks.size(); ++i)
        if (P.blocks[i].second >= bytes && P.blocks[i].second <= 2 * bytes && (best == P.blocks.size() || P.blocks[i].second < P.blocks[best].second)) best = i;
    if (best == P.blocks.size()) return nullptr;
    void* q = P.blocks[best].first; P.held -= P.blocks[best].second;
    P.blocks[best] = P.blocks.back(); P.blocks.pop_back();
    return q;
}
bool host_pool_give(void* q, size_t bytes) {
    const size_t limit = host_pool_limit();
    HostPool& P = g_host_pool;
    std::lock_guard<std::mutex> g(P.mu);
    if (P.held + bytes > limit) return false;
    P.blocks.emplace_back(q, bytes); P.held += bytes;
    return true;
}

// ---- janitor thread
namespace {
struct Janitor {
    std::thread th; std::mutex mu;
    ~Janitor() { if (th.joinable()) th.join(); }
    void run(std::shared_ptr<void> garbage) {
        std::lock_guard<std::mutex> g(mu);
        if (th.joinable()) th.join();
        th = std::thread([garbage]() mutable { garbage.reset(); });
    }
    // a set-up about to take big host arrays: what the previous one released should be in the pool by then (a back-to-back second set-up
    // that overtook the janitor took fresh pages instead -- download 43-74 ms instead of 4-6)
    void wait() { std::lock_guard<std::mutex> g(mu); if (th.joinable()) th.join(); }
};
Janitor g_janitor;
}  // namespace
void janitor_run(std::shared_ptr<void> garbage) { g_janitor.run(std::move(garbage)); }
void janitor_wait() { g_janitor.wait(); }

// ---- Ilup(p >= 1), ilup.rs:77-167
// IKJ elimination with level-of-fill bookkeeping on sparse rows instead of the reference's dense n x n `level` / `a_work` arrays: an entry
// that the dense code never touches is (0.0, usize::MAX) here too.  Whether an entry takes part depends on its VALUE too (`!= 0.0` tests at
// :106, :117, :129), so pattern and values are computed together, row by row.  Row i only needs the finished rows j < i that appear in its
// working row (original entries and fill): rows are dealt out to the host's cores in blocks, round-robin, every thread walks its blocks in
// ascending order and waits on a row's "finished" flag before it uses another thread's pivot row.  The lowest unfinished row never waits
// for an unfinished one, so somebody always makes progress.  The same operations on the same operands in the same order as the one-thread
// loop: same bits.
namespace {
struct IlupU { int32_t c; double v; uint64_t lev; };                       // a nonzero a_work[j][k], k > j, of a finished row
struct IlupE { int32_t c; double v; };                                     // a kept entry of L or U
template <class T>
struct Arena {                                                             // append-only; what has been handed out never moves (other threads read it)
    std::vector<hvec<T>> chunks; size_t used = 0, cap = 0;                 // (8 MiB chunks: huge pages, HostAlloc)
    // room for `cnt` elements with its pages already mapped: a thread that takes page faults (or maps a new chunk) in the middle of the row
    // pipeline holds up every thread behind it
    void reserve_mapped(size_t cnt) {
        cap = std::max<size_t>(cnt, ((size_t)8 << 20) / sizeof(T)); chunks.emplace_back(); chunks.back().resize(cap); used = 0;
        char* q = reinterpret_cast<char*>(chunks.back().data()); const size_t bytes = cap * sizeof(T);
#ifdef MADV_POPULATE_WRITE
        if (madvise(reinterpret_cast<void*>((reinterpret_cast<uintptr_t>(q) + 4095) & ~(uintptr_t)4095), bytes > 8192 ? (bytes - 4096) & ~(size_t)4095 : 0, MADV_POPULATE_WRITE) == 0) return;
#endif
        for (size_t x = 0; x < bytes; x += 4096) q[x] = 0;
    }
    T* take(size_t cnt) {
        if (used + cnt > cap) { cap = std::max<size_t>(cnt, ((size_t)8 << 20) / sizeof(T)); chunks.emplace_back(); chunks.back().resize(cap); used = 0; }
        T* p = chunks.back().data() + used; used += cnt; return p;
    }
};
struct WEnt { int32_t c; double v; uint64_t lev; };
struct RowOut { const IlupU* u = nullptr; const IlupE* l = nullptr; const IlupE* k = nullptr; int32_t nu = 0, nl = 0, nk = 0; };
}  // namespace

int host_ilup_rows(int64_t n, const int64_t* rp, const int32_t* col, const double* val, int fill, const IlupOptions& opt,
                   FlatRows& le, FlatRows& ue, hvec<double>& dg, long long* zero_pivot_col, std::shared_ptr<void>* scratch) {
    const bool verbose = opt.verbose;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto t_phase = now();
    auto lap = [&](const char* what) { if (verbose) { fprintf(stderr, "[kryst ilup] %s %.0f ms\n", what, std::chrono::duration<double, std::milli>(now() - t_phase).count()); t_phase = now(); } };
    const uint64_t UMAX = ~0ull;
    struct Bundle { std::vector<Arena<IlupU>> u; std::vector<Arena<IlupE>> l, k; hvec<RowOut> rows; hvec<double> udiag; };
    auto bundle = std::make_shared<Bundle>();
    hvec<RowOut>& rows = bundle->rows;                                     // finished rows: upper part (for later rows), kept L and U entries
    hvec<double>& udiag = bundle->udiag;                                   // a_work[j][j] of a finished row
    rows.resize((size_t)n); udiag.resize((size_t)n); dg.resize((size_t)n); // dg: the kept diagonal
    par_rows(n, [&](int64_t lo, int64_t hi) { for (int64_t i = lo; i < hi; ++i) { rows[(size_t)i] = RowOut(); udiag[(size_t)i] = 0.0; dg[(size_t)i] = 1.0; } });
    std::atomic<long long> bad_row{-1};                                    // lowest row i whose elimination met a zero u_jj ...
    std::vector<long long> bad_col;                                        // ... and that j, per thread
    const unsigned hw = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
    const unsigned T = (unsigned)std::max(1, opt.threads > 0 ? opt.threads : (n < 4096 ? 1 : (int)hw));
    // Blocks of 2 048 rows (measured at 128^3, 16 threads: 512 rows 197 ms, 1 024 110, 2 048 89, 4 096 90): on a grid with lines of 128 rows that is 16 lines, and row i's pivot rows i - 1, i - Ni (+ 1), i - Ni Nj (+ 1, + Ni)
    // are the thread's own except along the block's first line -- a pivot row finished by another core costs a few cache-line
    // transfers (0.2 - 1 us each on the two-socket hosts of the GPU boxes; blocks of 8 rows, tried first, were 3 x SLOWER than one thread).
    // A thread publishes its finished rows 16 at a time (and before it waits itself): one flag byte per row, so a consumer walking a
    // line behind its producer takes the flags' cache line once per batch, not once per row.
    const int64_t B = std::max<int64_t>(1, opt.block), nblocks = (n + B - 1) / B;
    std::unique_ptr<std::atomic<uint8_t>[]> done(new std::atomic<uint8_t>[(size_t)n + 64]);
    par_rows(n, [&](int64_t lo, int64_t hi) { for (int64_t r = lo; r < hi; ++r) done[(size_t)r].store(0, std::memory_order_relaxed); });
    bad_col.assign(T, -1);
    std::vector<double> waited(T, 0.0), busy(T, 0.0); std::vector<long long> waits(T, 0);   // (verbose: where a thread's time went)
    std::vector<long long> bad_at(T, -1);
    bundle->u.resize(T); bundle->l.resize(T); bundle->k.resize(T);
    std::vector<Arena<IlupU>>& arena_u = bundle->u; std::vector<Arena<IlupE>>&arena_l = bundle->l, &arena_k = bundle->k;   // (alive until the gather below)
    std::atomic<unsigned> warm{0};
    std::atomic<int> oom{0};
    auto worker = [&](unsigned tid) {
        const auto tw0 = std::chrono::steady_clock::now();
        struct Stop { const std::chrono::steady_clock::time_point t0; double* out; ~Stop() { *out = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); } } stop{tw0, &busy[tid]};
        Arena<IlupU>& au = arena_u[tid]; Arena<IlupE>& al = arena_l[tid]; Arena<IlupE>& ak = arena_k[tid];
        std::vector<WEnt> W; std::vector<IlupE> L, Kp; std::vector<IlupU> U;
        W.reserve(256); L.reserve(256); Kp.reserve(256); U.reserve(256);
        // every thread's first block waits for the one before it, so a slow start (a core waking up, the first pages of its arenas) would be paid
        // T times in a row: touch the first chunk of each arena, then start together
        if (T > 1) {
            // (an estimate of this thread's share: (fill + 1) times the operator's entries above / below the diagonal, and a tenth on top;
            // a thread that runs out continues in 8 MiB chunks)
            const size_t share = (size_t)((double)std::max<int64_t>(0, rp[(size_t)n] - n) * 0.5 * (double)(fill + 1) * 1.1 / (double)T) + 4096;
            au.reserve_mapped(share); al.reserve_mapped(share); ak.reserve_mapped(share);
            warm.fetch_add(1);
            for (unsigned spin = 0; warm.load(std::memory_order_acquire) < T && oom.load(std::memory_order_relaxed) == 0; ++spin) if ((spin & 1023) == 1023) std::this_thread::yield();
        }
        for (int64_t b = tid; b < nblocks; b += T) {
            const bool trace = verbose && opt.trace && b < 6 * (int64_t)T && (tid < 3 || tid == T - 1);
            const double tb0 = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tw0).count(), wb0 = waited[tid];
            int64_t published = b * B;                                      // rows [b B, published) of this block carry their flag
            auto publish = [&](int64_t upto) { for (; published < upto; ++published) done[(size_t)published].store(1, std::memory_order_release); };
            for (int64_t i = b * B; i < std::min(n, (b + 1) * B); ++i) {
                { const long long br = bad_row.load(std::memory_order_relaxed); if ((br >= 0 && br < i) || oom.load(std::memory_order_relaxed)) return; }   // the reference stopped before this row
                W.clear(); L.clear(); Kp.clear(); U.clear();
                for (int64_t k = rp[i]; k < rp[i + 1]; ++k)
                    if (col[k] < n) W.push_back(WEnt{col[k], val[k], val[k] != 0.0 ? 0ull : UMAX});   // ilup.rs:88-101 (halo columns dropped)
                std::sort(W.begin(), W.end(), [](const WEnt& x, const WEnt& y) { return x.c < y.c; });   // (a block's local numbering is ascending already)
                // The working row is a small SORTED array; iterating it by index while inserting fill entries BEHIND the cursor visits
                // exactly the columns `for j in 0..i` would (an inserted column is > j).
                for (size_t p = 0; p < W.size() && W[p].c < i; ++p) {                              // :104 `for j in 0..i`
                    const int32_t j = W[p].c;
                    const double ejv = W[p].v; const uint64_t ejl = W[p].lev;
                    if (!(ejv != 0.0 && ejl <= (uint64_t)fill)) continue;                          // :106
                    if (j < b * B && done[(size_t)j].load(std::memory_order_acquire) == 0) {      // row j is somebody else's and still under way
                        publish(i);                                                                // (nobody waits for what this thread has finished)
                        const auto w0 = std::chrono::steady_clock::now();
                        ++waits[tid];
                        for (unsigned spin = 0; done[(size_t)j].load(std::memory_order_acquire) == 0; ++spin) {
                            const long long br = bad_row.load(std::memory_order_relaxed);
                            if ((br >= 0 && br < i) || oom.load(std::memory_order_relaxed)) return;
                            if ((spin & 1023) == 1023) std::this_thread::yield();
                        }
                        waited[tid] += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - w0).count();
                    }
                    const double u_jj = udiag[(size_t)j];
                    if (u_jj == 0.0) {                                                             // :108-110
                        long long cur = bad_row.load();
                        while ((cur < 0 || i < cur) && !bad_row.compare_exchange_weak(cur, (long long)i)) {}
                        if (bad_at[tid] < 0 || i < bad_at[tid]) { bad_at[tid] = i; bad_col[tid] = j; }
                        return;
                    }
                    const double lij = ejv / u_jj;                                                 // :112
                    L.push_back(IlupE{j, lij});
                    size_t q = p + 1;                                                              // both lists ascend: one merge pass per pivot row
                    const IlupU* uj = rows[(size_t)j].u;
                    for (int32_t t = 0; t < rows[(size_t)j].nu; ++t) {                             // :116 `for k in (j+1)..n`
                        uint64_t nl = ejl;                                                         // saturating adds (:118)
                        nl = (nl > UMAX - uj[t].lev) ? UMAX : nl + uj[t].lev;
                        nl = (nl == UMAX) ? UMAX : nl + 1;
                        if (nl <= (uint64_t)fill) {
                            const int32_t k = uj[t].c;
                            while (q < W.size() && W[q].c < k) ++q;
                            if (q == W.size() || W[q].c != k) W.insert(W.begin() + (std::ptrdiff_t)q, WEnt{k, 0.0, UMAX});
                            W[q].v = W[q].v - lij * uj[t].v;                                       // :121
                            if (nl < W[q].lev) W[q].lev = nl;                                      // :122
                        }
                    }
                }
                for (const WEnt& e : W) {
                    if (e.c < i) continue;
                    if (e.c == i) udiag[(size_t)i] = e.v;
                    if (e.v != 0.0 && e.lev <= (uint64_t)fill) {                                   // :129-134
                        if (e.c == i) dg[(size_t)i] = e.v;
                        else Kp.push_back(IlupE{e.c, e.v});
                    }
                    if (e.c > i && e.v != 0.0) U.push_back(IlupU{e.c, e.v, e.lev});
                }
                RowOut& r = rows[(size_t)i];
                r.nu = (int32_t)U.size(); r.nl = (int32_t)L.size(); r.nk = (int32_t)Kp.size();
                if (r.nu) { IlupU* d = au.take(U.size()); std::copy(U.begin(), U.end(), d); r.u = d; }
                if (r.nl) { auto* d = al.take(L.size()); std::copy(L.begin(), L.end(), d); r.l = d; }
                if (r.nk) { auto* d = ak.take(Kp.size()); std::copy(Kp.begin(), Kp.end(), d); r.k = d; }
                if (((i + 1) & 15) == 0) publish(i + 1);
            }
            publish(std::min(n, (b + 1) * B));
            if (trace) fprintf(stderr, "[kryst ilup]     thread %u block %lld: %.3f .. %.3f ms, waited %.3f\n", tid, (long long)b, tb0,
                               std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tw0).count(), waited[tid] - wb0);
        }
    };
    // a worker that runs out of memory (HostAlloc throws) must not take the process down from a thread function, nor leave the others
    // spinning on a row that will never be published: it raises `oom`, everybody leaves, the caller reports it
    auto guarded = [&](unsigned t) {
        try { worker(t); } catch (const std::bad_alloc&) { oom.store(1); } catch (...) { oom.store(2); }
    };
    {
        // the workers exchange finished rows through the caches: keep them on neighbouring cores (one group of 16 consecutive CPU numbers
        // around the caller's: one socket, two L3 domains on the GPU boxes' hosts) -- cpu_group = 0 leaves the placement to the OS
        const int group = opt.cpu_group;
        const int cpu0 = sched_getcpu();
        auto placed = [&](unsigned t) {
            if (group > 0 && cpu0 >= 0 && T > 1) {
                cpu_set_t set; CPU_ZERO(&set);
                const int base = cpu0 / group * group;
                for (int c = base; c < base + group && c < CPU_SETSIZE; ++c) CPU_SET(c, &set);
                (void)pthread_setaffinity_np(pthread_self(), sizeof set, &set);     // (refused: the OS places the thread)
            }
            guarded(t);
        };
        if (T == 1) guarded(0);
        else {
            std::vector<std::thread> th;
            for (unsigned t = 0; t < T; ++t) th.emplace_back(placed, t);
            for (auto& t : th) t.join();
        }
    }
    if (scratch) *scratch = bundle;
    if (oom.load()) return 2;
    if (bad_row.load() >= 0) {
        const long long i = bad_row.load();
        long long j = -1;
        for (unsigned t = 0; t < T; ++t) if (bad_at[t] == i) j = bad_col[t];
        if (zero_pivot_col) *zero_pivot_col = j;
        return 1;
    }
    lap("elimination");
    if (verbose) for (unsigned t = 0; t < T; ++t) fprintf(stderr, "[kryst ilup]   thread %u: %.0f ms, of which %.0f ms in %lld waits for another thread's rows\n", t, busy[t], waited[t], waits[t]);
    le.ptr.assign((size_t)n + 1, 0); ue.ptr.assign((size_t)n + 1, 0);
    for (int64_t i = 0; i < n; ++i) { le.ptr[(size_t)i + 1] = le.ptr[(size_t)i] + rows[(size_t)i].nl; ue.ptr[(size_t)i + 1] = ue.ptr[(size_t)i] + rows[(size_t)i].nk; }
    le.col.resize((size_t)le.ptr[(size_t)n]); le.val.resize(le.col.size()); ue.col.resize((size_t)ue.ptr[(size_t)n]); ue.val.resize(ue.col.size());
    par_rows(n, [&](int64_t lo, int64_t hi) {
        for (int64_t i = lo; i < hi; ++i) {
            const RowOut& r = rows[(size_t)i];
            for (int32_t t = 0; t < r.nl; ++t) { le.col[(size_t)le.ptr[(size_t)i] + t] = r.l[t].c; le.val[(size_t)le.ptr[(size_t)i] + t] = r.l[t].v; }
            for (int32_t t = 0; t < r.nk; ++t) { ue.col[(size_t)ue.ptr[(size_t)i] + t] = r.k[t].c; ue.val[(size_t)ue.ptr[(size_t)i] + t] = r.k[t].v; }
        }
    });
    lap("kept entries gathered");
    return 0;
}

// ---- Ilut, ilut.rs:80-150: rows are independent (nothing is eliminated): counted and written by the host's cores, straight into flat arrays
void host_ilut_rows(int64_t n, const int64_t* rp, const int32_t* col, const double* val, int fill, double droptol,
                    FlatRows& le, FlatRows& ue, hvec<double>& dg, int threads) {
    dg.resize((size_t)n);
    par_rows(n, [&](int64_t lo, int64_t hi) { for (int64_t i = lo; i < hi; ++i) dg[(size_t)i] = 1.0; }, threads);
    le.ptr.assign((size_t)n + 1, 0); ue.ptr.assign((size_t)n + 1, 0);
    auto kept_row = [&](int64_t i, std::vector<std::pair<int32_t, double>>& row) {
        row.clear();
        for (int64_t k = rp[i]; k < rp[i + 1]; ++k)
            if (col[k] < n && val[k] != 0.0 && std::fabs(val[k]) >= droptol) row.push_back({col[k], val[k]});     // :88-95
        if ((int64_t)row.size() > fill) {                                                                        // :97-100
            std::stable_sort(row.begin(), row.end(), [](const std::pair<int32_t, double>& x, const std::pair<int32_t, double>& y) {
                return std::fabs(x.second) > std::fabs(y.second); });
            row.resize((size_t)fill);
        }
    };
    par_rows(n, [&](int64_t lo, int64_t hi) {
        std::vector<std::pair<int32_t, double>> row;
        for (int64_t i = lo; i < hi; ++i) {
            kept_row(i, row);
            int64_t nl = 0, nu = 0;
            for (auto& e : row) { if (e.first < i) ++nl; else if (e.first > i) ++nu; }
            le.ptr[(size_t)i + 1] = nl; ue.ptr[(size_t)i + 1] = nu;
        }
    }, threads);
    for (int64_t i = 0; i < n; ++i) { le.ptr[(size_t)i + 1] += le.ptr[(size_t)i]; ue.ptr[(size_t)i + 1] += ue.ptr[(size_t)i]; }
    le.col.resize((size_t)le.ptr[(size_t)n]); le.val.resize(le.col.size()); ue.col.resize((size_t)ue.ptr[(size_t)n]); ue.val.resize(ue.col.size());
    par_rows(n, [&](int64_t lo, int64_t hi) {
        std::vector<std::pair<int32_t, double>> row;
        for (int64_t i = lo; i < hi; ++i) {
            kept_row(i, row);
            int64_t pl = le.ptr[(size_t)i], pu = ue.ptr[(size_t)i];
            bool have_d = false;
            for (auto& e : row) {                                                                                // :104-112
                if (e.first < i) { le.col[(size_t)pl] = e.first; le.val[(size_t)pl] = e.second; ++pl; }
                else if (e.first > i) { ue.col[(size_t)pu] = e.first; ue.val[(size_t)pu] = e.second; ++pu; }
                else if (!have_d) { dg[(size_t)i] = e.second; have_d = true; }                                   // :143-144
            }
        }
    }, threads);
}

// ---- dependency levels of a triangular factor
int32_t host_levels(int64_t n, const int64_t* ptr, const int32_t* col, bool forward, int32_t* lvl) {
    int32_t nl = 0;
    auto level_of = [&](int64_t i) {
        int32_t l = 0;
        for (int64_t k = ptr[i]; k < ptr[i + 1]; ++k) l = std::max(l, lvl[col[k]] + 1);
        lvl[i] = l; nl = std::max(nl, l + 1);
    };
    if (forward) for (int64_t i = 0; i < n; ++i) level_of(i);
    else for (int64_t i = n - 1; i >= 0; --i) level_of(i);
    return nl;
}

}  // namespace kr
