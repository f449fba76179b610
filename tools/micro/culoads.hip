// Micro-benchmark: what ONE compute unit can stream.  Each workgroup (W waves) reads its own contiguous region, every wave with
// D 16-byte-per-lane loads (1 KiB each) in flight, issued and counted by hand (in-order return: wait until only D - G remain, use
// the oldest G).  Prints GB/s per workgroup for 1, 32 and 256 workgroups (one per CU at most) -- the rate a wavefront triangular
// solve's leading blocks can be fed at, as opposed to the whole chip's HBM rate.
// build: hipcc --offload-arch=gfx950 -O3 tools/micro/culoads.hip -o tools/micro/culoads
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned int u4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) u4 gu4;

template <int D>
__global__ __launch_bounds__(512) void stream(const char* base, long long bytes_per_wave, long long* out, unsigned* sink) {
    const int l = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const gu4* p = (const gu4*)(base + ((long long)blockIdx.x * nw + w) * bytes_per_wave) + l;
    const long long n = bytes_per_wave / 1024;                   // loads per wave
    u4 buf[D];
    unsigned acc = 0;
    const long long w0 = wall_clock64();
#pragma unroll
    for (int d = 0; d < D; ++d) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(buf[d]) : "v"(p + (long long)d * 64) : "memory");
    for (long long i = D; i < n; i += D) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            asm volatile("s_waitcnt vmcnt(%0)" : : "n"(D - 1) : "memory");
            asm volatile("" : "+v"(buf[d]));
            acc ^= buf[d].x ^ buf[d].w;
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(buf[d]) : "v"(p + (i + d) * 64) : "memory");
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int d = 0; d < D; ++d) { asm volatile("" : "+v"(buf[d])); acc ^= buf[d].y; }
    const long long w1 = wall_clock64();
    if (l == 0) out[blockIdx.x * nw + w] = w1 - w0;
    if (acc == 0x12345678u) sink[0] = acc;
}

template <int D> void run(const char* base, long long* out, unsigned* sink, int blocks, int waves, long long bpw) {
    for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL((stream<D>), dim3(blocks), dim3(64 * waves), 0, 0, base, bpw, out, sink); hipDeviceSynchronize(); }
    static long long h[256 * 8];
    hipMemcpy(h, out, sizeof(long long) * blocks * waves, hipMemcpyDeviceToHost);
    long long mx = 0; for (int i = 0; i < blocks * waves; ++i) mx = h[i] > mx ? h[i] : mx;
    const double us = mx / 100.0;
    printf("blocks %3d waves %d loads in flight per wave %2d (%3d KiB per CU): %7.1f GB/s per workgroup, %7.1f GB/s total, implied latency %5.2f us\n",
           blocks, waves, D, D * waves, bpw * waves / us / 1e3, bpw * waves * (double)blocks / us / 1e3, D * waves * 1024.0 / (bpw * waves / us) );
    fflush(stdout);
}

int main() {
    const long long bpw = 4ll << 20;                              // 4 MiB per wave
    char* base; long long* out; unsigned* sink;
    hipMalloc(&base, (size_t)bpw * 8 * 256 + (1 << 20)); hipMemset(base, 1, (size_t)bpw * 8 * 256 + (1 << 20));   // (+ slack: the last round of a wave reads up to D - 1 loads past its region)
    hipMalloc(&out, sizeof(long long) * 256 * 8); hipMalloc(&sink, 64);
    for (int blocks : {1, 32, 256})
        for (int waves : {1, 4, 8}) {
            run<8>(base, out, sink, blocks, waves, bpw);
            run<16>(base, out, sink, blocks, waves, bpw);
            run<32>(base, out, sink, blocks, waves, bpw);
            run<48>(base, out, sink, blocks, waves, bpw);
        }
    return 0;
}
