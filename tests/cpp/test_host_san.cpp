// The host-side concurrency of the library under ThreadSanitizer / AddressSanitizer + UBSan, on the CPU alone:
//   * the Ilup(p) row pipeline (host_factor.cpp: per-row flag bytes, round-robin blocks, worker arenas) and the parallel Ilut rows at 1, 3 and 16
//     threads with blocks of 1 ... 2 048 rows, compared with the oracle's dense restatements (kro_ilup_build / kro_ilut_build, compiled in
//     without OpenMP) entry by entry -- including WHICH zero pivot is reported, with the zero pivot in every position of a block;
//   * the level scheduler;
//   * the host block pool and the janitor thread (blocks released on one thread while the next set-up allocates);
//   * the counted shared mappings of inter-process handles (ipc_table.h) through fake open / close calls, hammered by eight threads.
// Built and run by `make -C kryst_amd/csrc san SAN=thread` (or SAN=address,undefined); exit code 0 and no sanitizer report = pass.
#include "../../kryst_amd/csrc/host_factor.h"
#include "../../kryst_amd/csrc/ipc_table.h"
extern "C" {
#include "../../oracle/kryst_oracle.h"
}
#include <cstdio>
#include <cstring>
#include <random>

using namespace kr;

static int g_fail = 0;
#define CHECK(cond, ...) do { if (!(cond)) { fprintf(stderr, "FAIL %s:%d: ", __FILE__, __LINE__); fprintf(stderr, __VA_ARGS__); fprintf(stderr, "\n"); ++g_fail; } } while (0)

struct Csr { int64_t n; std::vector<int64_t> rp, ci64; std::vector<int32_t> ci; std::vector<double> va; };

static Csr stencil7(int N) {                       // 7-point operator, mildly unsymmetric (what Ilup(1) sees of a convection-diffusion grid)
    Csr a; a.n = (int64_t)N * N * N; a.rp.push_back(0);
    for (int k = 0; k < N; ++k) for (int j = 0; j < N; ++j) for (int i = 0; i < N; ++i) {
        const int64_t r = i + (int64_t)N * (j + (int64_t)N * k);
        auto put = [&](int64_t c, double v) { a.ci.push_back((int32_t)c); a.ci64.push_back(c); a.va.push_back(v); };
        if (k > 0) put(r - (int64_t)N * N, -1.25);
        if (j > 0) put(r - N, -1.5);
        if (i > 0) put(r - 1, -2.0);
        put(r, 7.75);
        if (i < N - 1) put(r + 1, -1.0);
        if (j < N - 1) put(r + N, -1.0);
        if (k < N - 1) put(r + (int64_t)N * N, -1.0);
        a.rp.push_back((int64_t)a.ci.size());
    }
    return a;
}

static Csr random_band(int64_t n, int per_row, int64_t reach, unsigned seed, int64_t zero_diag_row = -1) {
    std::mt19937_64 rng(seed);
    std::uniform_real_distribution<double> val(-1.0, 1.0);
    Csr a; a.n = n; a.rp.push_back(0);
    for (int64_t i = 0; i < n; ++i) {
        std::vector<int64_t> cols{i};
        for (int t = 0; t < per_row; ++t) { int64_t c = i + (int64_t)(rng() % (2 * reach + 1)) - reach; c = std::max<int64_t>(0, std::min(n - 1, c)); cols.push_back(c); }
        std::sort(cols.begin(), cols.end()); cols.erase(std::unique(cols.begin(), cols.end()), cols.end());
        double sum = 0.0; std::vector<double> v(cols.size());
        for (size_t t = 0; t < cols.size(); ++t) if (cols[t] != i) { v[t] = val(rng); sum += std::fabs(v[t]); }
        for (size_t t = 0; t < cols.size(); ++t) {
            if (i == zero_diag_row && cols[t] < i) continue;                                 // the zero-pivot row has no lower entries ...
            double x = cols[t] == i ? (i == zero_diag_row ? 0.0 : sum + 1.0) : v[t];        // ... and a stored zero on its diagonal
            a.ci.push_back((int32_t)cols[t]); a.ci64.push_back(cols[t]); a.va.push_back(x);
        }
        a.rp.push_back((int64_t)a.ci.size());
    }
    return a;
}

// the oracle's TRIROWS factors (diagonal inside U's rows) against the host entry points' (strictly-lower L, strictly-upper U, kept diagonal)
static void compare(const char* what, const Csr& a, const kro_trirows_t& t, const FlatRows& le, const FlatRows& ue, const hvec<double>& dg) {
    const int64_t n = a.n;
    CHECK((int64_t)le.ptr.size() == n + 1 && (int64_t)ue.ptr.size() == n + 1 && (int64_t)dg.size() >= n, "%s: sizes", what);
    for (int64_t i = 0; i < n && g_fail == 0; ++i) {
        CHECK(t.l_ptr[i + 1] - t.l_ptr[i] == le.ptr[(size_t)i + 1] - le.ptr[(size_t)i], "%s: L row %lld length", what, (long long)i);
        for (int64_t k = t.l_ptr[i], q = le.ptr[(size_t)i]; k < t.l_ptr[i + 1] && g_fail == 0; ++k, ++q)
            CHECK(t.l_col[k] == le.col[(size_t)q] && t.l_val[k] == le.val[(size_t)q], "%s: L row %lld entry %lld", what, (long long)i, (long long)(k - t.l_ptr[i]));
        int64_t q = ue.ptr[(size_t)i]; double d = 1.0; bool seen = false;
        for (int64_t k = t.u_ptr[i]; k < t.u_ptr[i + 1] && g_fail == 0; ++k) {
            if (t.u_col[k] == i) { if (!seen) { d = t.u_val[k]; seen = true; } continue; }
            CHECK(q < ue.ptr[(size_t)i + 1] && t.u_col[k] == ue.col[(size_t)q] && t.u_val[k] == ue.val[(size_t)q], "%s: U row %lld", what, (long long)i);
            ++q;
        }
        CHECK(q == ue.ptr[(size_t)i + 1], "%s: U row %lld length", what, (long long)i);
        CHECK(d == dg[(size_t)i], "%s: diagonal of row %lld", what, (long long)i);
    }
}

static void ilup_cases() {
    const int threads[] = {1, 3, 16};
    const int64_t blocks[] = {1, 2, 7, 64, 2048};
    std::vector<std::pair<const char*, Csr>> mats;
    mats.emplace_back("stencil 7^3", stencil7(7));
    mats.emplace_back("random band 400", random_band(400, 6, 30, 1));
    for (auto& nm : mats) {
        const Csr& a = nm.second;
        const kro_csr_t oa{a.n, a.n, a.rp.data(), a.ci64.data(), a.va.data()};
        for (int fill = 1; fill <= 3; ++fill) {
            kro_trirows_t ref; memset(&ref, 0, sizeof ref);
            CHECK(kro_ilup_build(&oa, fill, &ref) == KRO_OK, "oracle ilup");
            for (int T : threads) for (int64_t B : blocks) {
                IlupOptions opt; opt.threads = T; opt.block = B; opt.cpu_group = 0;
                FlatRows le, ue; hvec<double> dg; long long zc = -1; std::shared_ptr<void> scratch;
                const int rc = host_ilup_rows(a.n, a.rp.data(), a.ci.data(), a.va.data(), fill, opt, le, ue, dg, &zc, &scratch);
                CHECK(rc == 0, "%s fill %d threads %d block %lld: rc %d", nm.first, fill, T, (long long)B, rc);
                char what[128]; snprintf(what, sizeof what, "%s ilup(%d) threads %d block %lld", nm.first, fill, T, (long long)B);
                if (rc == 0) compare(what, a, ref, le, ue, dg);
                janitor_run(scratch);                                   // released on the janitor thread while the next case allocates
            }
            kro_trirows_free(&ref);
        }
    }
    // a zero pivot in every position of a block of 8 rows (and the following block): the LOWEST row that meets one decides, whichever thread ran ahead
    for (int64_t j = 0; j < 18; ++j) {
        const Csr a = random_band(96, 5, 9, 77, j);
        const kro_csr_t oa{a.n, a.n, a.rp.data(), a.ci64.data(), a.va.data()};
        kro_trirows_t ref; memset(&ref, 0, sizeof ref);
        const int32_t orc = kro_ilup_build(&oa, 1, &ref);
        bool used = false;                                              // does any later row have an entry (or fill) in column j?  the oracle knows
        used = orc == KRO_SOLVE_ERROR;
        for (int T : threads) {
            IlupOptions opt; opt.threads = T; opt.block = 8; opt.cpu_group = 0;
            FlatRows le, ue; hvec<double> dg; long long zc = -1;
            const int rc = host_ilup_rows(a.n, a.rp.data(), a.ci.data(), a.va.data(), 1, opt, le, ue, dg, &zc, nullptr);
            CHECK((rc == 1) == used, "zero pivot at %lld, threads %d: rc %d, oracle %d", (long long)j, T, rc, orc);
            if (rc == 1) CHECK(zc == j, "zero pivot at %lld, threads %d: reported column %lld", (long long)j, T, zc);
        }
        if (orc == KRO_OK) kro_trirows_free(&ref);
    }
}

static void ilut_cases() {
    const Csr a = random_band(500, 7, 40, 3);
    const kro_csr_t oa{a.n, a.n, a.rp.data(), a.ci64.data(), a.va.data()};
    const std::pair<int, double> prm[] = {{2, 0.0}, {4, 1e-3}, {3, 0.3}, {50, 0.0}};
    for (auto& p : prm) {
        kro_trirows_t ref; memset(&ref, 0, sizeof ref);
        CHECK(kro_ilut_build(&oa, p.first, p.second, &ref) == KRO_OK, "oracle ilut");
        for (int T : {1, 3, 16}) {
            FlatRows le, ue; hvec<double> dg;
            host_ilut_rows(a.n, a.rp.data(), a.ci.data(), a.va.data(), p.first, p.second, le, ue, dg, T);
            char what[96]; snprintf(what, sizeof what, "ilut(%d, %g) threads %d", p.first, p.second, T);
            compare(what, a, ref, le, ue, dg);
        }
        kro_trirows_free(&ref);
    }
}

static void level_cases() {
    const Csr a = stencil7(6);
    std::vector<int64_t> lp{0}; std::vector<int32_t> lc;
    for (int64_t i = 0; i < a.n; ++i) { for (int64_t k = a.rp[i]; k < a.rp[i + 1]; ++k) if (a.ci[k] < i) lc.push_back(a.ci[k]); lp.push_back((int64_t)lc.size()); }
    std::vector<int32_t> lvl((size_t)a.n);
    const int32_t nl = host_levels(a.n, lp.data(), lc.data(), true, lvl.data());
    CHECK(nl == 3 * 6 - 2, "levels of a 6^3 grid: %d", nl);
    for (int64_t r = 0; r < a.n; ++r) CHECK(lvl[(size_t)r] == (int32_t)(r % 6 + (r / 6) % 6 + r / 36), "level of row %lld", (long long)r);
}

// ---- host block pool + janitor: big blocks released on the janitor thread while other threads allocate and free
static void pool_cases() {
    std::vector<std::thread> th;
    for (int t = 0; t < 4; ++t)
        th.emplace_back([t] {
            for (int r = 0; r < 6; ++r) {
                auto v = std::make_shared<hvec<double>>();
                v->resize(((size_t)5 << 20) / 8 + (size_t)t * 1024);     // >= 4 MiB: a pooled, huge-page-hinted block
                (*v)[0] = 1.0; (*v)[v->size() - 1] = 2.0;
                if (r % 2 == 0) janitor_run(v); else v.reset();
            }
        });
    for (auto& t : th) t.join();
    janitor_wait();
}

// ---- counted shared mappings through fakes
static std::atomic<int> g_fake_opens{0}, g_fake_closes{0}, g_fake_live{0};
struct FakeOps {
    static int open(void** ptr, const SharedMappingKey& key) {
        if (key[0] == 'X') return 17;                                    // a handle the "runtime" refuses
        *ptr = ::operator new(16);
        ++g_fake_opens; ++g_fake_live;
        return 0;
    }
    static void close(void* ptr) { ::operator delete(ptr); ++g_fake_closes; --g_fake_live; }
};
static void ipc_table_cases() {
    SharedMappings<FakeOps> table;
    std::vector<std::thread> th;
    std::atomic<int> bad{0};
    for (int t = 0; t < 8; ++t)
        th.emplace_back([&, t] {
            for (int round = 0; round < 200; ++round) {
                void* p[4];
                for (int h = 0; h < 4; ++h) {                            // four handles, shared by all eight threads, on "device" t % 2
                    SharedMappingKey key{}; key[0] = (char)('a' + h); const int dev = t % 2; memcpy(key.data() + 64, &dev, 4);
                    if (table.open(&p[h], key) != 0) ++bad;
                }
                for (int h = 3; h >= 0; --h) table.close(p[h]);
                SharedMappingKey refused{}; refused[0] = 'X';
                void* q = nullptr;
                if (table.open(&q, refused) != 17) ++bad;
            }
        });
    for (auto& t : th) t.join();
    CHECK(bad.load() == 0, "shared mappings: %d unexpected results", bad.load());
    CHECK(g_fake_live.load() == 0 && table.size() == 0, "shared mappings: %d mappings left open, %zu table entries", g_fake_live.load(), table.size());
    CHECK(g_fake_opens.load() == g_fake_closes.load() && g_fake_opens.load() >= 8, "shared mappings: %d opens, %d closes", g_fake_opens.load(), g_fake_closes.load());
    // a mapping that is NOT the table's goes straight to close
    void* foreign = ::operator new(16); ++g_fake_live;
    table.close(foreign);
    CHECK(g_fake_live.load() == 0, "foreign close");
}

int main() {
    ilup_cases();
    ilut_cases();
    level_cases();
    pool_cases();
    ipc_table_cases();
    janitor_wait();
    if (g_fail) { fprintf(stderr, "test_host_san: %d check(s) failed\n", g_fail); return 1; }
    printf("test_host_san ok: Ilup row pipeline (1 / 3 / 16 threads x blocks 1..2048 x fill 1..3, zero pivot in 18 positions), Ilut rows, level scheduler, "
           "host pool + janitor, counted shared mappings\n");
    return 0;
}
