// Does hipMemset on the NULL stream order with work on a hipStreamNonBlocking stream, and has it run when it returns?
// (The question behind the round-2 fault of the multi-rank test: ilu.hip zeroed a preconditioner's argument block with
// hipMemset and wrote it right afterwards from a kernel on the context's non-blocking compute stream.)
//
//   hipcc -O2 --offload-arch=gfx950 tools/micro/nullstream_memset.hip -o tools/micro/nullstream_memset && tools/micro/nullstream_memset
//
// Test 1: time of a 4 GiB hipMemset call against the time until the device is idle: a call that returns in microseconds has not
//         run the fill.
// Test 2: a ~2 ms spin kernel is queued on the null stream, then hipMemset(word, 0) on the null stream, then a kernel on a
//         non-blocking stream stores 1 into the word, stream sync, device sync.  word == 0 at the end means the fill ran AFTER
//         the store: the hazard.  (word == 1: the fill had completed, or at least ordered, before the call returned.)
// Test 3: the same with the fix (hipMemsetAsync on the non-blocking stream itself): must always end with 1.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void spin_kernel(long long cycles, int* sink) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < cycles) {}
    if (sink) *sink = 1;
}
__global__ void store_kernel(long long* w) { w[0] = 1; w[1] = 1; w[2] = 1; w[3] = 1; }

static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main() {
    CK(hipSetDevice(0));
    {
        const size_t bytes = 4ull << 30;
        char* p = nullptr;
        CK(hipMalloc(&p, bytes));
        CK(hipMemset(p, 1, bytes)); CK(hipDeviceSynchronize());
        const double t0 = now_ms();
        CK(hipMemset(p, 0, bytes));
        const double t1 = now_ms();
        CK(hipDeviceSynchronize());
        const double t2 = now_ms();
        printf("test1: hipMemset(4 GiB) returned after %.3f ms, device idle after another %.3f ms -> %s\n", t1 - t0, t2 - t1,
               (t2 - t1) > 5 * (t1 - t0) ? "ASYNCHRONOUS to the host" : "host-synchronous");
        CK(hipFree(p));
    }
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    long long* w = nullptr; int* sink = nullptr;
    CK(hipMalloc(&w, 64)); CK(hipMalloc(&sink, 4));
    const long long spin = 200000;           // wall_clock64 ticks at 100 MHz: 2 ms
    for (int variant = 0; variant < 2; ++variant) {
        int late = 0;
        const int reps = 50;
        for (int r = 0; r < reps; ++r) {
            CK(hipMemsetAsync(w, 0xff, 64, s)); CK(hipStreamSynchronize(s)); CK(hipDeviceSynchronize());
            hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(1), 0, 0, spin, sink);          // null stream busy for ~2 ms
            if (variant == 0) CK(hipMemset(w, 0, 32));                                     // round 2: fill on the null stream
            else { CK(hipMemsetAsync(w, 0, 32, s)); CK(hipStreamSynchronize(s)); }         // round 3: fill on the compute stream, waited for
            hipLaunchKernelGGL(store_kernel, dim3(1), dim3(1), 0, s, w);
            CK(hipStreamSynchronize(s));
            CK(hipDeviceSynchronize());
            long long h[4];
            CK(hipMemcpy(h, w, 32, hipMemcpyDeviceToHost));
            if (h[0] != 1) ++late;
        }
        printf("test%d: %s: the fill landed AFTER the compute stream's store in %d of %d trials\n", 2 + variant,
               variant == 0 ? "hipMemset on the null stream, store on a non-blocking stream" : "hipMemsetAsync on the same stream + sync", late, reps);
    }
    return 0;
}
