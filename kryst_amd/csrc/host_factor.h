// Host-side factorisations of the ILU family, free of any device call: the row pipeline of Ilup::new(p).setup (ilup.rs:77-134), the
// independent rows of Ilut::setup (ilut.rs:80-117) and the dependency levels of a triangular factor (the level scheduler of the general
// triangular solve).  ilu.hip calls them between the download of the operator's rows and the upload of the factors; the C ABI exports them
// on plain host arrays (kryst_host_ilup / kryst_host_ilut / kryst_host_levels, include/kryst_hip.h) so that a CPU-only caller -- and the
// sanitizer tier, `make -C kryst_amd/csrc san SAN=thread` -- can run exactly the code the GPU set-up runs.  No HIP header in here.
#pragma once
#include <algorithm>
#include <atomic>
#include <cstdint>
#include <cstdlib>
#include <memory>
#include <mutex>
#include <new>
#include <thread>
#include <utility>
#include <vector>
#include <sys/mman.h>

namespace kr {

// host-side loops over independent rows, split over the box's cores (setup only); `threads` <= 0: up to 16 of the hardware's
template <class F>
static void par_rows(int64_t n, F body, int threads = 0) {
    const unsigned hw = threads > 0 ? (unsigned)threads : std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
    if ((n < (1 << 16) && threads <= 0) || hw == 1 || n < 2) { body((int64_t)0, n); return; }
    std::vector<std::thread> th;
    const int64_t per = (n + hw - 1) / hw;
    for (unsigned t = 0; t < hw; ++t) {
        const int64_t lo = std::min<int64_t>(n, per * t), hi = std::min<int64_t>(n, lo + per);
        if (lo < hi) th.emplace_back([=] { body(lo, hi); });
    }
    for (auto& t : th) t.join();
}

// Host arrays of the setup paths (hundreds of MB on a 128^3 operator): elements are NOT value-initialised (a std::vector zeroes what the
// next loop overwrites, on one thread), and blocks of 4 MiB and more are 2 MiB-aligned with a huge-page hint -- first-touch page faults
// and the unmapping at the end were a quarter of the Ilup(1) setup (round 4).
// Big blocks are not given back to the OS but kept (KRYST_HOST_POOL_MB, default 2048; 0: off) for the next set-up: mapping and unmapping ~1 GB per
// Ilup set-up at 128^3 was a third of its time, and the unmapping of one set-up got in the way of the page faults of the next.
void* host_pool_take(size_t bytes);                // host_factor.cpp
bool host_pool_give(void* q, size_t bytes);
template <class T>
struct HostAlloc {
    using value_type = T;
    HostAlloc() = default;
    template <class U> HostAlloc(const HostAlloc<U>&) {}
    static size_t rounded(size_t bytes) { const size_t big = (size_t)1 << 21; return (bytes + big - 1) / big * big; }
    T* allocate(size_t cnt) {
        const size_t bytes = cnt * sizeof(T);
        void* q = nullptr;
        if (bytes >= ((size_t)1 << 22)) {
            if ((q = host_pool_take(rounded(bytes)))) return static_cast<T*>(q);
            if (posix_memalign(&q, (size_t)1 << 21, rounded(bytes)) != 0) throw std::bad_alloc();
            (void)madvise(q, rounded(bytes), MADV_HUGEPAGE);
        } else if (!(q = std::malloc(std::max<size_t>(bytes, 1)))) throw std::bad_alloc();
        return static_cast<T*>(q);
    }
    void deallocate(T* q, size_t cnt) {
        const size_t bytes = cnt * sizeof(T);
        if (bytes >= ((size_t)1 << 22) && host_pool_give(q, rounded(bytes))) return;
        std::free(q);
    }
    template <class U, class... A> void construct(U* q, A&&... a) {
        if constexpr (sizeof...(A) == 0) ::new ((void*)q) U; else ::new ((void*)q) U(std::forward<A>(a)...);
    }
    template <class U> bool operator==(const HostAlloc<U>&) const { return true; }
    template <class U> bool operator!=(const HostAlloc<U>&) const { return false; }
};
template <class T> using hvec = std::vector<T, HostAlloc<T>>;

// releases big host arrays off the caller's thread; at most one release in flight, joined before the next and when the library goes
void janitor_run(std::shared_ptr<void> garbage);   // the last owner's destructor runs on the janitor thread
void janitor_wait();                               // what the previous set-up released is in the pool after this

// kept entries of a triangular factor, row by row in STORED order (flat CSR: a vector per row costs 2n heap blocks)
struct FlatRows {
    hvec<int64_t> ptr; hvec<int32_t> col; hvec<double> val;
    int64_t len(int64_t i) const { return ptr[(size_t)i + 1] - ptr[(size_t)i]; }
};

struct IlupOptions {
    int threads = 0;            // <= 0: min(16, hardware threads), 1 below 4 096 rows
    int64_t block = 2048;       // rows per block of the round-robin row pipeline
    int cpu_group = 16;         // workers pinned to the group of this many consecutive CPUs around the caller's (0: placement left to the OS)
    bool verbose = false, trace = false;
};
// Ilup::new(fill).setup on the rows (rp, col, val) of an n x n block (columns >= n -- halo slots of a row-partitioned operator -- are
// dropped): kept strictly-lower entries with their multipliers (le), kept strictly-upper entries (ue), the kept diagonal (dg, 1.0 where
// none is kept).  Returns 0, or 1 when the elimination met a zero u_jj (ilup.rs:108-110): *zero_pivot_col then holds that j of the LOWEST
// row that met one -- what the one-thread loop would have reported.  *scratch receives what the caller should hand to janitor_run.
int host_ilup_rows(int64_t n, const int64_t* rp, const int32_t* col, const double* val, int fill, const IlupOptions& opt,
                   FlatRows& le, FlatRows& ue, hvec<double>& dg, long long* zero_pivot_col, std::shared_ptr<void>* scratch);
// Ilut::new(fill, droptol).setup (ilut.rs:80-117: no elimination -- drop by magnitude, keep the `fill` largest of a row, split at the diagonal)
void host_ilut_rows(int64_t n, const int64_t* rp, const int32_t* col, const double* val, int fill, double droptol,
                    FlatRows& le, FlatRows& ue, hvec<double>& dg, int threads = 0);
// dependency levels of a triangular factor given by its strictly-lower (forward) or strictly-upper (!forward) rows: lvl[i] = 1 + the highest
// level among the rows i depends on (0 when it depends on none).  Returns the number of levels.
int32_t host_levels(int64_t n, const int64_t* ptr, const int32_t* col, bool forward, int32_t* lvl);

}  // namespace kr
