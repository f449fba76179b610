// Micro-benchmark: the solving wave's step of tri_quad.h WITHOUT any waiting -- four waves per workgroup, each wave per step: two
// 8-byte LDS reads from another wave's ring (broadcast patterns of the west / south operands), one 64-bit lane permute + one
// DPP shift of its previous result, the zero-coefficient selects, three mul + three subtract (fp64, unfused), one 64-lane ring
// write and a lane-0 counter write; every other step a 16-byte stage read.  Prints ns per step for 1 and 256 workgroups and for
// 1 / 2 / 4 active waves: the floor of the step as written, against which the kernel's 150 ns is read.
// build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/micro/quadstep.hip -o tools/micro/quadstep
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ double bperm(int idx, double v) {
    int lo = __builtin_amdgcn_ds_bpermute(idx, __double2loint(v)), hi = __builtin_amdgcn_ds_bpermute(idx, __double2hiint(v));
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double shr1(double v) {
    int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x111, 0xf, 0xf, false), hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x111, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
template <int MODE>
__global__ __launch_bounds__(256) void k(int steps, int nwaves, long long* out, double* sink) {
    __shared__ double ring[4 * 32 * 64];
    __shared__ __attribute__((aligned(16))) double stage[8 * 256];
    __shared__ int prog[4];
    const int l = threadIdx.x & 63, q = threadIdx.x >> 6, jl = l & 7, kl = l >> 3;
    for (int i = threadIdx.x; i < 4 * 32 * 64; i += 256) ring[i] = 0.001 * i;
    for (int i = threadIdx.x; i < 8 * 256; i += 256) stage[i] = 1.0 + i;
    __syncthreads();
    if (q >= nwaves) return;
    const double* wp = ring + ((q + 3) & 3) * 32 * 64 + 7 + 8 * kl;
    const double* sp = ring + ((q + 2) & 3) * 32 * 64 + 56 + jl;
    double* mine = ring + q * 32 * 64 + l;
    const int idx8 = max(l - 8, 0) * 4;
    double y = 1.0 + l;
    double a1 = 0.25, a2 = 0.125, a3 = 0.0625;
    if (l == 63) a3 = 0.0;
    const long long w0 = wall_clock64();
    for (int t = 0; t < steps; t += 8) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int tt = t + u;
            double wv = 0.5, sv = 0.25;
            if (MODE & 1) { wv = wp[((tt + 7) & 31) * 64]; sv = sp[((tt + 7) & 31) * 64]; }
            double yj = shr1(y), yk = bperm(idx8, y);
            if (jl == 0) yj = wv;
            if (kl == 0) yk = sv;
            yj = a2 != 0.0 ? yj : 0.0; yk = a3 != 0.0 ? yk : 0.0;
            const double yi = a1 != 0.0 ? y : 0.0;
            double s = 3.0;
            if (MODE & 2) { const v2 rp = *reinterpret_cast<const v2*>(stage + ((u >> 1) * 256 + 64 * q + l) * 2); s = (u & 1) ? rp.y : rp.x; }
            s = s - a3 * yk; s = s - a2 * yj; s = s - a1 * yi;
            y = s;
            if (MODE & 4) {
                mine[(tt & 31) * 64] = s;
                if (MODE & 8) {                                   // variants of the counter write
                    if (MODE & 16) { asm volatile("" ::: "memory"); __hip_atomic_store(&prog[q], tt + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); asm volatile("" ::: "memory"); }   // all lanes
                    else if (MODE & 32) { if (u & 1) { asm volatile("" ::: "memory"); if (l == 0) __hip_atomic_store(&prog[q], tt + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); asm volatile("" ::: "memory"); } }   // every other step
                    // else: no counter
                } else {
                    asm volatile("" ::: "memory");
                    if (l == 0) __hip_atomic_store(&prog[q], tt + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    asm volatile("" ::: "memory");
                }
            }
        }
    }
    const long long w1 = wall_clock64();
    if (l == 0 && q == 0 && blockIdx.x == 0) out[0] = w1 - w0;
    if (y == 12345.678) sink[0] = y;
}
template <int MODE> void run(const char* name, long long* out, double* sink, int blocks, int nwaves) {
    const int steps = 1 << 15;
    for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(256), 0, 0, steps, nwaves, out, sink); hipDeviceSynchronize(); }
    long long h; hipMemcpy(&h, out, 8, hipMemcpyDeviceToHost);
    printf("%-52s blocks %3d waves %d: %6.1f ns per step\n", name, blocks, nwaves, h * 10.0 / steps); fflush(stdout);
}
int main() {
    long long* out; double* sink; hipMalloc(&out, 8); hipMalloc(&sink, 8);
    for (int blocks : {1, 256})
        for (int nw : {1, 4}) {
            run<0>("chain + permute + dpp + selects", out, sink, blocks, nw);
            run<1>("+ west / south ring reads", out, sink, blocks, nw);
            run<3>("+ stage read (16 B per two steps)", out, sink, blocks, nw);
            run<7>("+ ring write and counter (the whole step)", out, sink, blocks, nw);
            run<15>("  ring write, no counter", out, sink, blocks, nw);
            run<31>("  ring write, counter written by all lanes", out, sink, blocks, nw);
            run<47>("  ring write, lane-0 counter every other step", out, sink, blocks, nw);
        }
    return 0;
}
