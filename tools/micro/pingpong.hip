// Micro-benchmark: store -> load visibility latency between two workgroups on MI355X, by scope and XCD placement.
// Two chosen blocks of a small grid play ping-pong through two flags; reports ns per one-way hop and both blocks' XCC ids.
// build: hipcc --offload-arch=gfx950 -O3 tools/micro/pingpong.hip -o tools/micro/pingpong
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
// SCOPE 1: L2-coherent accesses (sc0: miss in the CU's L1, served by the XCD's L2) -- only coherent between CUs of one XCD
template <int SCOPE> __device__ __forceinline__ unsigned long long ld(unsigned long long* p) {
    if constexpr (SCOPE == 0) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned long long v;
    asm volatile("global_load_dwordx2 %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}
template <int SCOPE> __device__ __forceinline__ void st(unsigned long long* p, unsigned long long v) {
    if constexpr (SCOPE == 0) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return; }
    asm volatile("global_store_dwordx2 %0, %1, off sc0" : : "v"(p), "v"(v) : "memory");
}
template <int SCOPE>
__global__ void pp(unsigned long long* flags, int a, int b, int iters, int* xcc, long long* cycles) {
    const int me = blockIdx.x;
    if (threadIdx.x != 0 || (me != a && me != b)) return;
    unsigned id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
    xcc[me == a ? 0 : 1] = (int)(id & 0xf);
    unsigned long long* mine = flags + (me == a ? 0 : 32);       // 256 B apart
    unsigned long long* theirs = flags + (me == a ? 32 : 0);
    const long long t0 = wall_clock64();
    for (int it = 1; it <= iters; ++it) {
        if (me == a) {
            st<SCOPE>(mine, (unsigned long long)it);
            for (int bud = 1 << 12; bud > 0 && ld<SCOPE>(theirs) < (unsigned long long)it; --bud) {}
        } else {
            for (int bud = 1 << 12; bud > 0 && ld<SCOPE>(theirs) < (unsigned long long)it; --bud) {}
            st<SCOPE>(mine, (unsigned long long)it);
        }
    }
    if (me == a) *cycles = wall_clock64() - t0;
}
int main() {
    unsigned long long* flags; int* xcc; long long* cyc;
    hipMalloc(&flags, 4096); hipMalloc(&xcc, 8); hipMalloc(&cyc, 8);
    const int iters = 500;
    struct { const char* name; int scope; int a, b; } cases[] = {
        {"agent     same XCD (blocks 0, 8)", 0, 0, 8}, {"agent     other XCD (blocks 0, 1)", 0, 0, 1},
        {"workgroup same XCD (blocks 0, 8)", 1, 0, 8}, {"agent     same XCD (blocks 0,256)", 0, 0, 256},
        {"workgroup same XCD (blocks 0,256)", 1, 0, 256}, {"agent     other XCD (blocks 0,259)", 0, 0, 259},
    };
    for (auto& c : cases) {
        for (int rep = 0; rep < 2; ++rep) {
            hipMemset(flags, 0, 4096);
            if (c.scope == 0) hipLaunchKernelGGL(pp<0>, dim3(264), dim3(64), 0, 0, flags, c.a, c.b, iters, xcc, cyc);
            else hipLaunchKernelGGL(pp<1>, dim3(264), dim3(64), 0, 0, flags, c.a, c.b, iters, xcc, cyc);
            if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 1; }
        }
        int hx[2]; long long hc;
        hipMemcpy(hx, xcc, 8, hipMemcpyDeviceToHost); hipMemcpy(&hc, cyc, 8, hipMemcpyDeviceToHost);
        // wall_clock64 ticks at 100 MHz
        printf("%-36s xcc %d/%d  %.0f ns per one-way hop\n", c.name, hx[0], hx[1], (double)hc * 10.0 / (2.0 * iters));
        fflush(stdout);
    }
    return 0;
}
