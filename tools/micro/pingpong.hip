// Micro-benchmark: store -> load visibility latency between two workgroups on MI355X, by scope and XCD placement.
// Two chosen blocks of a small grid play ping-pong through two flags; reports ns per one-way hop and both blocks' XCC ids.
// build: hipcc --offload-arch=gfx950 -O3 tools/micro/pingpong.hip -o tools/micro/pingpong
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
// SCOPE 1: sc0 accesses.  Measured (round 3): they are served by the CU's own L1 -- two workgroups exchange through them only when they
// share a CU (188 ns); on different CUs of one XCD the poller never sees the store (budget exhausted).  There is no "L2-scope" access:
// bypassing L1 means agent scope (sc1), 460 ns between CUs of one XCD, 580 ns across XCDs.
// SCOPE 2: agent-scope (write-through) STORES with L2-coherent (sc0) LOADS: what a consumer can do when it knows its producer
// sits on the same XCD, while the producer stores for any consumer
template <int SCOPE> __device__ __forceinline__ unsigned long long ld(unsigned long long* p) {
    if constexpr (SCOPE == 0) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned long long v;
    asm volatile("global_load_dwordx2 %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}
template <int SCOPE> __device__ __forceinline__ void st(unsigned long long* p, unsigned long long v) {
    if constexpr (SCOPE == 0 || SCOPE == 2) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return; }
    asm volatile("global_store_dwordx2 %0, %1, off sc0" : : "v"(p), "v"(v) : "memory");
}
template <int SCOPE>
__global__ void pp(unsigned long long* flags, int a, int b, int iters, int* xcc, long long* cycles) {
    const int me = blockIdx.x;
    if (threadIdx.x != 0 || (me != a && me != b)) return;
    unsigned id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
    unsigned hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    xcc[me == a ? 0 : 1] = (int)(id & 0xf) | (int)(((hw >> 8) & 0xf) << 8) | (int)(((hw >> 13) & 0x7) << 16) | (int)(((hw >> 12) & 1) << 12);   // xcc | cu << 8 | sh << 12 | se << 16
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
        {"agent st + sc0 ld same XCD (0,256)", 2, 0, 256}, {"agent st + sc0 ld same XCD (0,16)", 2, 0, 16}, {"agent st + sc0 ld same XCD (0,128)", 2, 0, 128},
        {"workgroup same XCD (0,16)", 1, 0, 16}, {"workgroup same XCD (0,24)", 1, 0, 24}, {"workgroup same XCD (0,32)", 1, 0, 32}, {"workgroup same XCD (0,64)", 1, 0, 64},
        {"workgroup same XCD (0,128)", 1, 0, 128}, {"workgroup same XCD (8,72)", 1, 8, 72}, {"workgroup same XCD (0,248)", 1, 0, 248}, {"workgroup same XCD (16,40)", 1, 16, 40},
    };
    for (auto& c : cases) {
        for (int rep = 0; rep < 2; ++rep) {
            hipMemset(flags, 0, 4096);
            if (c.scope == 0) hipLaunchKernelGGL(pp<0>, dim3(264), dim3(64), 0, 0, flags, c.a, c.b, iters, xcc, cyc);
            else if (c.scope == 2) hipLaunchKernelGGL(pp<2>, dim3(264), dim3(64), 0, 0, flags, c.a, c.b, iters, xcc, cyc);
            else hipLaunchKernelGGL(pp<1>, dim3(264), dim3(64), 0, 0, flags, c.a, c.b, iters, xcc, cyc);
            if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 1; }
        }
        int hx[2]; long long hc;
        hipMemcpy(hx, xcc, 8, hipMemcpyDeviceToHost); hipMemcpy(&hc, cyc, 8, hipMemcpyDeviceToHost);
        // wall_clock64 ticks at 100 MHz
        printf("%-36s xcc %d/%d se %d/%d sh %d/%d cu %d/%d  %.0f ns per one-way hop\n", c.name, hx[0] & 0xf, hx[1] & 0xf, (hx[0] >> 16) & 7, (hx[1] >> 16) & 7,
               (hx[0] >> 12) & 1, (hx[1] >> 12) & 1, (hx[0] >> 8) & 0xf, (hx[1] >> 8) & 0xf, (double)hc * 10.0 / (2.0 * iters));
        fflush(stdout);
    }
    return 0;
}
