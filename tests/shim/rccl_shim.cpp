// TEST INFRASTRUCTURE: a stand-in for librccl.so.1 that lets SEVERAL RANKS SHARE ONE GPU (real RCCL refuses that), so the
// library's multi-rank code path -- communicator set-up from a broadcast id, the index-list exchange of
// kryst_csr_create_dist, the halo exchange launch sequence, the all-gather behind every inner product, the run-ahead
// termination rule -- can be executed end to end on a one-GPU test box (tests/test_gpu_z_multirank_shim.py).
// It implements exactly the nine entry points kryst_amd/csrc/dist.cpp binds, with the same prototypes as rccl.h, by staging
// through a POSIX shared-memory board: every call synchronises its stream, copies device -> board, meets the peers,
// copies board -> device.  Every copy is issued ON THE OPERATION'S OWN STREAM and waited for (like RCCL, which enqueues on
// the caller's stream): the library's streams are non-blocking, so a null-stream copy would not order with them.
// Selected with KRYST_RCCL_LIB=<path>; never used by the product otherwise.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fcntl.h>
#include <string>
#include <sys/mman.h>
#include <unistd.h>
#include <vector>
#include <chrono>
#include <thread>

namespace {
constexpr int MAXR = 8;                           // (ranks may also be threads of one process: all per-call state is thread_local)
constexpr size_t SLOT = 4u << 20;                 // bytes per (src,dst) mailbox
struct Board {
    std::atomic<int> arrived; std::atomic<int> generation;
    std::atomic<int> full[MAXR][MAXR];            // mailbox src -> dst holds a message
    std::atomic<long long> bytes[MAXR][MAXR];
    char pad[4096];
    char gather[MAXR][4096];                      // all-gather staging (<= 4 KiB per rank)
    char mail[MAXR][MAXR][SLOT];
};
struct Comm { Board* b; int rank, nranks; std::string name; };
struct Op { bool send; void* buf; size_t bytes; int peer; hipStream_t s; };
thread_local int g_group = 0;
thread_local std::vector<Op> g_ops;
thread_local Comm* g_comm = nullptr;

void spin(std::atomic<int>& a, int want) {
    long n = 0;
    while (a.load(std::memory_order_acquire) != want) {
        if (++n > 2000) std::this_thread::sleep_for(std::chrono::microseconds(50));
        if (n > 1200000) { fprintf(stderr, "rccl_shim: timeout waiting for a peer\n"); _exit(97); }      // 60 s
    }
}
void barrier(Comm* c) {
    Board* b = c->b;
    const int gen = b->generation.load(std::memory_order_acquire);
    if (b->arrived.fetch_add(1, std::memory_order_acq_rel) == c->nranks - 1) {
        b->arrived.store(0, std::memory_order_release);
        b->generation.store(gen + 1, std::memory_order_release);
    } else spin(b->generation, gen + 1);
}
// device <-> board on stream s, complete on return
hipError_t copy_on(void* dst, const void* src, size_t bytes, hipMemcpyKind kind, hipStream_t s) {
    if (bytes == 0) return hipSuccess;
    const hipError_t e = hipMemcpyAsync(dst, src, bytes, kind, s);
    return e != hipSuccess ? e : hipStreamSynchronize(s);
}
size_t dsize(ncclDataType_t t) { return (t == ncclFloat64 || t == ncclInt64 || t == ncclUint64) ? 8 : (t == ncclInt32 || t == ncclFloat32 || t == ncclUint32) ? 4 : 1; }

ncclResult_t run_ops(Comm* c, std::vector<Op>& ops) {
    for (auto& o : ops) if (hipStreamSynchronize(o.s) != hipSuccess) return ncclUnhandledCudaError;
    Board* b = c->b;
    for (auto& o : ops) if (o.send) {                                  // post every send first (mailboxes are private per pair)
        if (o.bytes > SLOT) { fprintf(stderr, "rccl_shim: message of %zu bytes exceeds the mailbox\n", o.bytes); return ncclInvalidArgument; }
        spin(b->full[c->rank][o.peer], 0);
        if (copy_on(b->mail[c->rank][o.peer], o.buf, o.bytes, hipMemcpyDeviceToHost, o.s) != hipSuccess) return ncclUnhandledCudaError;
        b->bytes[c->rank][o.peer].store((long long)o.bytes);
        b->full[c->rank][o.peer].store(1, std::memory_order_release);
    }
    for (auto& o : ops) if (!o.send) {
        spin(b->full[o.peer][c->rank], 1);
        if ((size_t)b->bytes[o.peer][c->rank].load() != o.bytes) { fprintf(stderr, "rccl_shim: send/recv size mismatch\n"); return ncclInvalidArgument; }
        if (copy_on(o.buf, b->mail[o.peer][c->rank], o.bytes, hipMemcpyHostToDevice, o.s) != hipSuccess) return ncclUnhandledCudaError;
        b->full[o.peer][c->rank].store(0, std::memory_order_release);
    }
    ops.clear();
    return ncclSuccess;
}
}  // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
    memset(id, 0, sizeof *id);
    snprintf(id->internal, sizeof id->internal, "kryst_shim_%d_%lld", (int)getpid(),
             (long long)std::chrono::steady_clock::now().time_since_epoch().count());
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId id, int rank) {
    if (nranks > MAXR) return ncclInvalidArgument;
    Comm* c = new Comm();
    c->rank = rank; c->nranks = nranks; c->name = std::string("/") + id.internal;
    int fd = -1;
    if (rank == 0) {
        fd = shm_open(c->name.c_str(), O_CREAT | O_RDWR, 0600);
        if (fd < 0 || ftruncate(fd, sizeof(Board)) != 0) return ncclSystemError;
    } else {
        for (int t = 0; t < 20000 && fd < 0; ++t) { fd = shm_open(c->name.c_str(), O_RDWR, 0600); if (fd < 0) usleep(1000); }
        if (fd < 0) return ncclSystemError;
        struct stat_dummy { } ;
        for (int t = 0; t < 20000; ++t) { off_t sz = lseek(fd, 0, SEEK_END); if (sz >= (off_t)sizeof(Board)) break; usleep(1000); }
    }
    void* p = mmap(nullptr, sizeof(Board), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) return ncclSystemError;
    c->b = static_cast<Board*>(p);                 // a fresh shm segment is zero-filled: counters start at 0
    *comm = reinterpret_cast<ncclComm_t>(c);
    barrier(c);
    if (rank == 0) shm_unlink(c->name.c_str());    // everybody has it mapped
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm) {
    Comm* c = reinterpret_cast<Comm*>(comm);
    if (c) { munmap(c->b, sizeof(Board)); delete c; }
    return ncclSuccess;
}

const char* ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "success" : "rccl_shim error"; }

ncclResult_t ncclGroupStart() { ++g_group; return ncclSuccess; }
ncclResult_t ncclGroupEnd() {
    if (--g_group > 0) return ncclSuccess;
    if (!g_comm) { g_ops.clear(); return ncclSuccess; }
    Comm* c = g_comm; g_comm = nullptr;
    return run_ops(c, g_ops);
}
ncclResult_t ncclSend(const void* buf, size_t count, ncclDataType_t t, int peer, ncclComm_t comm, hipStream_t s) {
    g_comm = reinterpret_cast<Comm*>(comm);
    g_ops.push_back(Op{true, const_cast<void*>(buf), count * dsize(t), peer, s});
    if (g_group == 0) { Comm* c = g_comm; g_comm = nullptr; return run_ops(c, g_ops); }
    return ncclSuccess;
}
ncclResult_t ncclRecv(void* buf, size_t count, ncclDataType_t t, int peer, ncclComm_t comm, hipStream_t s) {
    g_comm = reinterpret_cast<Comm*>(comm);
    g_ops.push_back(Op{false, buf, count * dsize(t), peer, s});
    if (g_group == 0) { Comm* c = g_comm; g_comm = nullptr; return run_ops(c, g_ops); }
    return ncclSuccess;
}

ncclResult_t ncclAllGather(const void* send, void* recv, size_t count, ncclDataType_t t, ncclComm_t comm, hipStream_t s) {
    Comm* c = reinterpret_cast<Comm*>(comm);
    const size_t bytes = count * dsize(t);
    if (bytes > 4096) return ncclInvalidArgument;
    if (hipStreamSynchronize(s) != hipSuccess) return ncclUnhandledCudaError;
    if (copy_on(c->b->gather[c->rank], send, bytes, hipMemcpyDeviceToHost, s) != hipSuccess) return ncclUnhandledCudaError;
    barrier(c);
    for (int p = 0; p < c->nranks; ++p)
        if (copy_on((char*)recv + p * bytes, c->b->gather[p], bytes, hipMemcpyHostToDevice, s) != hipSuccess) return ncclUnhandledCudaError;
    barrier(c);
    return ncclSuccess;
}

}  // extern "C"
