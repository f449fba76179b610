// Context management, error text, host-side problem generators.
#include "dist.h"
#include <cstdarg>
#include <map>
#include <mutex>
#include <unordered_map>

namespace kr {
std::atomic<uint32_t> g_env_epoch{1};
thread_local int g_env_frozen = 0;
static thread_local char g_err[1024] = "";
static thread_local int64_t g_err_row = -1;
void set_error_row(int64_t row) { g_err_row = row; }
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}
}  // namespace kr
namespace kr {
// ---- device block pool (common.h)
namespace {
struct DevPool {
    std::mutex mu;
    std::multimap<size_t, void*> free_blocks;         // size -> block, per device
    std::unordered_map<void*, size_t> live;           // blocks handed out that may come back (>= POOL_MIN bytes)
    size_t pooled = 0;
};
constexpr size_t POOL_MIN = (size_t)1 << 20;
DevPool g_pools[64];
size_t pool_cap_bytes() {
    static const size_t cap = [] { const char* e = getenv("KRYST_DEV_POOL_MB"); const long long mb = e ? atoll(e) : 65536; return mb <= 0 ? (size_t)0 : (size_t)mb << 20; }();
    return cap;
}
DevPool* pool_here(int* dev_out = nullptr) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); dev = 0; }
    if (dev_out) *dev_out = dev;
    return &g_pools[dev & 63];
}
}  // namespace
hipError_t pool_malloc_bytes(void** p, size_t bytes) {
    DevPool* P = pool_here();
    if (bytes >= POOL_MIN && pool_cap_bytes() > 0) {
        std::lock_guard<std::mutex> g(P->mu);
        auto it = P->free_blocks.find(bytes);
        if (it != P->free_blocks.end()) {
            *p = it->second; P->pooled -= bytes; P->free_blocks.erase(it); P->live[*p] = bytes;
            return hipSuccess;
        }
    }
    hipError_t e = hipMalloc(p, bytes);
    if (e != hipSuccess) {                             // no room: what the pool holds goes back to the driver, then once more
        (void)hipGetLastError();
        int dev = 0; (void)pool_here(&dev);
        if (pool_trim(dev) > 0) e = hipMalloc(p, bytes);
    }
    if (e == hipSuccess && bytes >= POOL_MIN && pool_cap_bytes() > 0) { std::lock_guard<std::mutex> g(P->mu); P->live[*p] = bytes; }
    return e;
}
hipError_t pool_free(void* p) {
    if (!p) return hipSuccess;
    DevPool* P = pool_here();
    size_t bytes = 0;
    {
        std::lock_guard<std::mutex> g(P->mu);
        auto it = P->live.find(p);
        if (it != P->live.end()) { bytes = it->second; P->live.erase(it); }
    }
    if (bytes == 0 || pool_cap_bytes() == 0) return hipFree(p);
    // hipFree waits for the device before it releases a block; a block that goes to the pool instead must not be handed to a new owner
    // under a kernel that still uses it either
    const hipError_t e = hipDeviceSynchronize();
    if (e != hipSuccess) { (void)hipGetLastError(); return hipFree(p); }
    // test hook (KRYST_DEV_POOL_POISON=1): a pooled block is handed out again UNCLEARED, so whatever a set-up reads it must have written itself --
    // the tests fill every block that enters the pool with 0xFF bytes (NaNs as doubles, -1 as indices)
    static const bool poison = [] { const char* e = getenv("KRYST_DEV_POOL_POISON"); return e && atoi(e) != 0; }();
    if (poison && (hipMemset(p, 0xFF, bytes) != hipSuccess || hipDeviceSynchronize() != hipSuccess)) (void)hipGetLastError();
    std::lock_guard<std::mutex> g(P->mu);
    if (P->pooled + bytes > pool_cap_bytes()) return hipFree(p);
    P->free_blocks.emplace(bytes, p); P->pooled += bytes;
    return hipSuccess;
}
size_t pool_trim(int device) {
    DevPool* P = &g_pools[device & 63];
    std::multimap<size_t, void*> blocks;
    size_t bytes = 0;
    { std::lock_guard<std::mutex> g(P->mu); blocks.swap(P->free_blocks); bytes = P->pooled; P->pooled = 0; }
    for (auto& b : blocks) (void)hipFree(b.second);
    (void)hipGetLastError();
    return bytes;
}

void phase_mark_slow(kryst_ctx_t ctx, int phase) {
    PhaseTimer* T = ctx->phase;
    hipEvent_t e = nullptr;
    if (!T->pool.empty()) { e = T->pool.back(); T->pool.pop_back(); }
    else if (hipEventCreate(&e) != hipSuccess) { (void)hipGetLastError(); return; }
    if (hipEventRecord(e, ctx->s_main) != hipSuccess) { (void)hipGetLastError(); T->pool.push_back(e); return; }
    T->marks.emplace_back(phase, e);
}
}  // namespace kr
using namespace kr;

static int32_t ctx_init(kryst_ctx_t ctx) {
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count == 0) {
        set_error("no HIP device available: libkryst_hip has no CPU fallback");
        return KRYST_ERR_HIP;
    }
    if (!(ctx->device >= 0 && ctx->device < count)) {
        set_error("bad argument: device id %d out of range (this process sees %d HIP device(s))", ctx->device, count);
        return KRYST_ERR_ARG;
    }
    KR_HIP(hipSetDevice(ctx->device));
    hipDeviceProp_t prop;
    KR_HIP(hipGetDeviceProperties(&prop, ctx->device));
    ctx->num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    KR_HIP(hipStreamCreateWithFlags(&ctx->s_main, hipStreamNonBlocking));
    KR_HIP(hipStreamCreateWithFlags(&ctx->s_comm, hipStreamNonBlocking));
    KR_HIP(hipEventCreateWithFlags(&ctx->ev_x_ready, hipEventDisableTiming));
    KR_HIP(hipEventCreateWithFlags(&ctx->ev_halo_done, hipEventDisableTiming));
    KR_HIP(hipEventCreate(&ctx->tm0));
    KR_HIP(hipEventCreate(&ctx->tm1));
    for (auto& e : ctx->ev_ring) KR_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    KR_HIP(hipMalloc(&ctx->d_scal, sizeof(double) * 4096));
    KR_HIP(hipMemsetAsync(ctx->d_scal, 0, sizeof(double) * 4096, ctx->s_main));
    KR_HIP(hipStreamSynchronize(ctx->s_main));
    KR_HIP(hipMalloc(&ctx->d_ticket, 64));
    KR_HIP(hipMemsetAsync(ctx->d_ticket, 0, 64, ctx->s_main));
    KR_HIP(hipStreamSynchronize(ctx->s_main));
    KR_HIP(hipMalloc(&ctx->d_gather, sizeof(double) * (size_t)(ctx->nranks + 1) * KR_MAXQ));
    KR_HIP(hipHostMalloc((void**)&ctx->h_prog, sizeof(HostProgress), hipHostMallocMapped));
    memset((void*)ctx->h_prog, 0, sizeof(HostProgress));
    KR_HIP(hipHostGetDevicePointer((void**)&ctx->d_prog, (void*)ctx->h_prog, 0));
    KR_HIP(hipHostMalloc((void**)&ctx->h_pinned, sizeof(double) * 4096, hipHostMallocDefault));
    return KRYST_OK;
}

extern "C" {

const char* kryst_hip_last_error(void) { return g_err; }
int64_t kryst_hip_last_error_row(void) { return g_err_row; }
int32_t kryst_hip_abi_version(void) { return 5; }
void kryst_reduce_spec(int32_t* T, int32_t* V, int32_t* F) {
    if (T) *T = KR_T;
    if (V) *V = KR_V;
    if (F) *F = KR_F;
}

int32_t kryst_device_count(int32_t* count) {
    KR_ARG(count, "device_count: out");
    int c = 0;
    if (hipGetDeviceCount(&c) != hipSuccess) { (void)hipGetLastError(); c = 0; }
    *count = c;
    return KRYST_OK;
}

int32_t kryst_ctx_create(int32_t device_id, kryst_ctx_t* out) {
    KR_ARG(out, "ctx_create: out");
    kryst_ctx_t ctx = new kryst_ctx_s();
    ctx->device = device_id;
    int32_t rc = ctx_init(ctx);
    if (rc != KRYST_OK) { kryst_ctx_destroy(ctx); return rc; }
    *out = ctx;
    return KRYST_OK;
}

int32_t kryst_comm_unique_id(void* out128) {
    KR_ARG(out128, "unique_id: out");
    return comm_unique_id(out128);
}

int32_t kryst_ctx_create_dist(int32_t device_id, int32_t rank, int32_t nranks, const void* uid, kryst_ctx_t* out) {
    KR_ARG(out && uid && nranks >= 1 && rank >= 0 && rank < nranks, "ctx_create_dist");
    kryst_ctx_t ctx = new kryst_ctx_s();
    ctx->device = device_id; ctx->rank = rank; ctx->nranks = nranks;
    int32_t rc = ctx_init(ctx);
    if (rc == KRYST_OK) rc = comm_init(ctx, uid);
    if (rc == KRYST_OK) {
        // The inner products cross the ranks through the hipIpc mailboxes whenever every rank can set them up AND one checked reduction
        // arrives intact on every rank (dist.cpp: ipc_reduce_setup -> ipc_reduce_selftest; agreed outcome); otherwise -- not an error -- the
        // RCCL all-gather stays.  KRYST_SCALAR_REDUCE=rccl: do not try (all ranks read the same environment).
        const char* e = getenv("KRYST_SCALAR_REDUCE");
        if (nranks > 1 && !(e && strcmp(e, "rccl") == 0)) { const int32_t r2 = ipc_reduce_setup(ctx); if (r2 != KRYST_OK && r2 != KRYST_UNSUPPORTED) rc = r2; }
    }
    if (rc != KRYST_OK) { kryst_ctx_destroy(ctx); return rc; }      // frees whatever ctx_init / comm_init got as far as creating
    *out = ctx;
    return KRYST_OK;
}

int32_t kryst_ctx_destroy(kryst_ctx_t ctx) {
    if (!ctx) return KRYST_OK;
    (void)hipSetDevice(ctx->device);
    if (ctx->s_main) (void)hipStreamSynchronize(ctx->s_main);
    if (ctx->s_comm) (void)hipStreamSynchronize(ctx->s_comm);
    comm_destroy(ctx);
    if (ctx->phase) { for (auto& m : ctx->phase->marks) (void)hipEventDestroy(m.second); for (auto& e : ctx->phase->pool) (void)hipEventDestroy(e); delete ctx->phase; }
    for (void* q : ctx->deferred_free) (void)hipFree(q);
    (void)pool_trim(ctx->device);                     // (per device: a second context on this device simply refills it)
    (void)hipFree(ctx->d_partials); (void)hipFree(ctx->d_chunks); (void)hipFree(ctx->d_ticket); (void)hipFree(ctx->d_scal); (void)hipFree(ctx->d_gather); (void)hipFree(ctx->arena);
    (void)hipHostFree((void*)ctx->h_prog); (void)hipHostFree(ctx->h_pinned);
    for (hipEvent_t e : {ctx->ev_x_ready, ctx->ev_halo_done, ctx->tm0, ctx->tm1}) if (e) (void)hipEventDestroy(e);
    for (auto& e : ctx->ev_ring) if (e) (void)hipEventDestroy(e);
    if (ctx->s_main) (void)hipStreamDestroy(ctx->s_main);
    if (ctx->s_comm) (void)hipStreamDestroy(ctx->s_comm);
    (void)hipGetLastError();
    delete ctx;
    return KRYST_OK;
}

int32_t kryst_ctx_synchronize(kryst_ctx_t ctx) {
    KR_ARG(ctx, "ctx");
    KR_HIP(hipSetDevice(ctx->device));
    KR_HIP(hipStreamSynchronize(ctx->s_comm));
    KR_HIP(hipStreamSynchronize(ctx->s_main));
    return KRYST_OK;
}

// Give back to the driver what the context keeps for reuse between calls: the device block pool of destroyed ILU-family preconditioners
// (common.h: pool_*; per device) and -- when no solve or stepping session is open -- the work-vector arena of the solvers.
int32_t kryst_ctx_trim(kryst_ctx_t ctx, int64_t* bytes_released) {
    KR_ARG(ctx, "ctx_trim");
    KR_HIP(hipSetDevice(ctx->device));
    KR_TRY(kryst_ctx_synchronize(ctx));
    size_t freed = pool_trim(ctx->device);
    if (ctx->active_ws == nullptr && ctx->arena_owner == nullptr && ctx->arena) {
        freed += ctx->arena_bytes;
        (void)hipFree(ctx->arena); ctx->arena = nullptr; ctx->arena_bytes = 0; ctx->arena_used = 0;
    }
    if (bytes_released) *bytes_released = (int64_t)freed;
    return KRYST_OK;
}

int32_t kryst_ctx_rank(kryst_ctx_t ctx, int32_t* rank, int32_t* nranks) {
    KR_ARG(ctx, "ctx");
    if (rank) *rank = ctx->rank;
    if (nranks) *nranks = ctx->nranks;
    return KRYST_OK;
}

int32_t kryst_comm_all_reduce(kryst_ctx_t ctx, double x, double* out) {
    KR_ARG(ctx && out, "all_reduce");
    if (ctx->nranks == 1) { *out = x; return KRYST_OK; }          // RayonComm::all_reduce: identity (rayon_comm.rs:76-78)
    KR_HIP(hipSetDevice(ctx->device));
    double* local = ctx->d_gather + (size_t)ctx->nranks * KR_MAXQ;
    KR_HIP(hipMemcpyAsync(local, &x, sizeof(double), hipMemcpyHostToDevice, ctx->s_main));
    KR_TRY(comm_all_gather(ctx, local, ctx->d_gather, 1));
    KR_HIP(hipMemcpyAsync(ctx->h_pinned, ctx->d_gather, sizeof(double) * ctx->nranks, hipMemcpyDeviceToHost, ctx->s_main));
    KR_HIP(hipStreamSynchronize(ctx->s_main));
    double total = ctx->h_pinned[0];
    for (int p = 1; p < ctx->nranks; ++p) total = total + ctx->h_pinned[p];   // rank order: same bits on every rank
    *out = total;
    return KRYST_OK;
}

// How the solvers' inner products cross the ranks: mode 0 = RCCL all-gather + rank-ordered fold, 1 = IPC mailboxes (the default when their
// set-up and test reduction succeed on every rank), -1 = query
// (dist.cpp: one launch per inner product, no collective).  COLLECTIVE: every rank of the context calls it with the same mode.
// KRYST_UNSUPPORTED (mode 1) when some rank cannot export / map a mailbox: every rank then stays on RCCL.  *active: the mode in use.
int32_t kryst_ctx_scalar_reduce(kryst_ctx_t ctx, int32_t mode, int32_t* active) {
    KR_ARG(ctx && (mode == 0 || mode == 1 || mode == -1), "ctx_scalar_reduce");
    if (mode == -1) { if (active) *active = ctx->ipc_on ? 1 : 0; return KRYST_OK; }      // query only
    KR_ARG(ctx->active_ws == nullptr, "ctx_scalar_reduce: a solve or stepping session is open on this context");
    int32_t rc = KRYST_OK;
    if (ctx->nranks > 1 || ctx->comm) {
        KR_HIP(hipSetDevice(ctx->device));
        KR_TRY(kryst_ctx_synchronize(ctx));
        if (mode == 1) rc = ipc_reduce_setup(ctx);
        else ctx->ipc_on = false;                 // (the mailboxes stay mapped: switching back costs nothing)
    }
    if (active) *active = ctx->ipc_on ? 1 : 0;
    return rc;
}

int32_t kryst_comm_barrier(kryst_ctx_t ctx) {
    KR_ARG(ctx, "ctx");
    KR_TRY(kryst_ctx_synchronize(ctx));
    double dummy;
    return kryst_comm_all_reduce(ctx, 0.0, &dummy);
}

int32_t kryst_phase_timing_begin(kryst_ctx_t ctx) {
    KR_ARG(ctx, "ctx");
    KR_HIP(hipSetDevice(ctx->device));
    if (!ctx->phase) ctx->phase = new PhaseTimer();
    for (auto& m : ctx->phase->marks) ctx->phase->pool.push_back(m.second);
    ctx->phase->marks.clear();
    phase_mark_slow(ctx, -1);
    return KRYST_OK;
}

int32_t kryst_phase_timing_end(kryst_ctx_t ctx, double* ms, int32_t count) {
    KR_ARG(ctx && ms && count >= KR_PH_COUNT, "phase_timing_end: need room for every phase");
    KR_ARG(ctx->phase, "phase_timing_end without phase_timing_begin");
    KR_HIP(hipSetDevice(ctx->device));
    KR_HIP(hipStreamSynchronize(ctx->s_comm));
    KR_HIP(hipStreamSynchronize(ctx->s_main));
    for (int i = 0; i < count; ++i) ms[i] = 0.0;
    PhaseTimer* T = ctx->phase;
    for (size_t k = 1; k < T->marks.size(); ++k) {
        float f = 0.f;
        if (hipEventElapsedTime(&f, T->marks[k - 1].second, T->marks[k].second) != hipSuccess) { (void)hipGetLastError(); continue; }
        const int ph = T->marks[k].first;
        if (ph >= 0 && ph < KR_PH_COUNT) ms[ph] += (double)f;
    }
    for (auto& m : T->marks) (void)hipEventDestroy(m.second);
    for (auto& e : T->pool) (void)hipEventDestroy(e);
    delete T;
    ctx->phase = nullptr;
    return KRYST_OK;
}

int32_t kryst_phase_count(void) { return KR_PH_COUNT; }

const char* kryst_phase_name(int32_t phase) {
    static const char* names[KR_PH_COUNT] = {"spmv", "halo_wait", "spmv_boundary", "reduce", "blas1", "pc", "blas1_residual", "blas1_direction", "blas1_xbatch"};
    return (phase >= 0 && phase < KR_PH_COUNT) ? names[phase] : "";
}

int32_t kryst_ctx_timer_start(kryst_ctx_t ctx) {
    KR_ARG(ctx, "ctx");
    KR_HIP(hipSetDevice(ctx->device));
    KR_HIP(hipEventRecord(ctx->tm0, ctx->s_main));
    return KRYST_OK;
}

int32_t kryst_ctx_timer_stop(kryst_ctx_t ctx, double* ms) {
    KR_ARG(ctx && ms, "timer_stop");
    KR_HIP(hipSetDevice(ctx->device));
    KR_HIP(hipEventRecord(ctx->tm1, ctx->s_main));
    KR_HIP(hipEventSynchronize(ctx->tm1));
    float f = 0.f;
    KR_HIP(hipEventElapsedTime(&f, ctx->tm0, ctx->tm1));
    *ms = (double)f;
    return KRYST_OK;
}

// ---- host-only synthetic problems: 7-point stencil, row = i + N*(j + N*k), ascending columns,
// homogeneous Dirichlet by truncation.
static void stencil_coefs(int kind, double c[7]) {
    // order of c: -N^2 (bottom), -N (south), -1 (west), 0 (diag), +1 (east), +N (north), +N^2 (top)
    if (kind == 0) { c[0] = c[1] = c[2] = c[4] = c[5] = c[6] = -1.0; c[3] = 6.0; }
    else if (kind == 1) { const double cx = 1.0, cy = 1.0, cz = 0.01;
        c[2] = c[4] = -cx; c[1] = c[5] = -cy; c[0] = c[6] = -cz; c[3] = 2.0 * (cx + cy + cz); }
    else { const double gx = 1.0, gy = 0.5, gz = 0.25;
        c[2] = -(1.0 + gx); c[4] = -1.0; c[1] = -(1.0 + gy); c[5] = -1.0; c[0] = -(1.0 + gz); c[6] = -1.0;
        c[3] = 6.0 + gx + gy + gz; }
}

// kind 3 ("varcoef", csr_create.hip): weight of the edge between rows r and r + {1, N, N^2}[d]
static inline double varcoef_weight(int64_t r, int d) {
    uint64_t z = 0xD1FFull + ((uint64_t)(3 * r + d) + 1) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z = z ^ (z >> 31);
    return 0.5 + (double)(z >> 11) * (1.0 / 9007199254740992.0);
}

int64_t kryst_host_stencil7(int32_t N, int32_t kind, int32_t k_lo, int32_t k_hi, int64_t* row_ptr, int64_t* col_idx,
                            double* vals) {
    if (N < 1 || k_lo < 0 || k_hi > N || k_lo > k_hi || kind < 0 || kind > 3) { set_error("stencil7: bad arguments"); return -1; }
    double c[7] = {0, 0, 0, 0, 0, 0, 0};
    if (kind < 3) stencil_coefs(kind, c);
    const int64_t N1 = N, N2 = N1 * N1;
    int64_t nnz = 0, lr = 0;
    if (row_ptr) row_ptr[0] = 0;
    for (int64_t k = k_lo; k < k_hi; ++k)
        for (int64_t j = 0; j < N1; ++j)
            for (int64_t i = 0; i < N1; ++i, ++lr) {
                const int64_t row = i + N1 * (j + N1 * k);
                const bool ok[7] = {k > 0, j > 0, i > 0, true, i < N1 - 1, j < N1 - 1, k < N1 - 1};
                const int64_t off[7] = {-N2, -N1, -1, 0, 1, N1, N2};
                if (kind == 3 && vals) {
                    const double w[7] = {ok[0] ? varcoef_weight(row - N2, 2) : 1.0, ok[1] ? varcoef_weight(row - N1, 1) : 1.0, ok[2] ? varcoef_weight(row - 1, 0) : 1.0, 0.0,
                                         ok[4] ? varcoef_weight(row, 0) : 1.0, ok[5] ? varcoef_weight(row, 1) : 1.0, ok[6] ? varcoef_weight(row, 2) : 1.0};
                    double dsum = 0.0;
                    for (int s = 0; s < 7; ++s) if (s != 3) { dsum = dsum + w[s]; c[s] = -w[s]; }
                    c[3] = dsum;
                }
                for (int s = 0; s < 7; ++s)
                    if (ok[s]) {
                        if (col_idx) col_idx[nnz] = row + off[s];
                        if (vals) vals[nnz] = c[s];
                        ++nnz;
                    }
                if (row_ptr) row_ptr[lr + 1] = nnz;
            }
    return nnz;
}

}  // extern "C"
