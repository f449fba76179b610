// Micro-benchmark: does the PLACEMENT of the plain-CSR SpMV's five arrays change how fast they stream?  The traffic of
// tools/micro/mixstream.hip (values 16 B per lane, indices 8 B per lane, dependent row pointers, x read once, y written nontemporally;
// one 512-row tile per workgroup) out of ONE allocation, the arrays at controlled byte offsets from each other: array a starts at
// its natural start (previous end rounded up to 2 MiB) + skew[a].  Each line: the skews, the median time of 5 launches.
// build: hipcc --offload-arch=gfx950 -O3 tools/micro/placement.hip -o tools/micro/placement
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned int u4 __attribute__((ext_vector_type(4)));
typedef unsigned int u2 __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(256) void mix(const char* val, const char* col, const int* rp, const char* x, char* y, int vb, unsigned* out) {
    const int t = threadIdx.x, l = t & 63, w = t >> 6;
    const long long q = blockIdx.x;
    unsigned acc = 0;
    const int p = rp[q * 512 + 2 * t];
    const int shift = __builtin_amdgcn_readfirstlane(p) & 16; acc ^= p;
    const int wv = vb / 4;
    const char* vbase = val + q * vb + (long long)w * wv + shift;
    const char* cbase = col + q * (vb / 2) + (long long)w * (wv / 2) + shift / 2;
    for (int off = 0; off < wv; off += 4 * 1024) {
        u4 v[4]; u2 c[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int o = min(off + u * 1024 + l * 16, wv - 16);
            v[u] = *reinterpret_cast<const u4*>(vbase + o);
            c[u] = *reinterpret_cast<const u2*>(cbase + o / 2);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) acc ^= v[u].x ^ v[u].y ^ v[u].z ^ v[u].w ^ c[u].x ^ c[u].y;
    }
    const u4 xv = *reinterpret_cast<const u4*>(x + q * 4096 + t * 16); acc ^= xv.x ^ xv.w;
    u4 yv; yv.x = acc; yv.y = t; yv.z = 0; yv.w = 1;
    __builtin_nontemporal_store(yv, reinterpret_cast<u4*>(y + q * 4096 + t * 16));
    if (acc == 0x12345678u) out[0] = acc;
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

int main(int argc, char** argv) {
    const long long ntiles = argc > 1 ? atoll(argv[1]) : 262144;          // 512^3
    const int vb = 28672;
    const size_t M2 = 2u << 20, slack = 64u << 20;
    const size_t sz[5] = {(size_t)ntiles * vb, (size_t)ntiles * vb / 2, (size_t)ntiles * 2048, (size_t)ntiles * 4096, (size_t)ntiles * 4096};
    size_t nat[5], total = 0;
    for (int a = 0; a < 5; ++a) { nat[a] = total; total += (sz[a] + slack + M2 - 1) / M2 * M2; }
    char* slab; unsigned* out;
    CK(hipMalloc(&slab, total + slack)); CK(hipMalloc(&out, 64));
    CK(hipMemset(slab, 0, total + slack)); CK(hipDeviceSynchronize());
    printf("slab %p, %zu bytes; skews (bytes) of val col rp x y -> ms\n", (void*)slab, total);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto run = [&](const size_t* skew) -> float {
        std::vector<float> ms;
        for (int r = 0; r < 6; ++r) {
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL(mix, dim3((unsigned)ntiles), dim3(256), 0, 0, slab + nat[0] + skew[0], slab + nat[1] + skew[1],
                               reinterpret_cast<const int*>(slab + nat[2] + skew[2]), slab + nat[3] + skew[3], slab + nat[4] + skew[4], vb, out);
            hipEventRecord(e1, 0); hipEventSynchronize(e1);
            float m; hipEventElapsedTime(&m, e0, e1);
            if (r) ms.push_back(m);
        }
        std::sort(ms.begin(), ms.end());
        return ms[2];
    };
    const size_t steps[] = {0, 256, 512, 1024, 2048, 4096, 8192, 16384, 32768, 65536, 131072, 262144, 524288, 1048576, 1048576 + 4096, 3145728 + 256, 5 * 1048576, 17 * 1048576 + 12288};
    for (int rep = 0; rep < 2; ++rep) {
        size_t z[5] = {0, 0, 0, 0, 0};
        printf("all zero: %.3f ms\n", run(z));
    }
    for (int a = 0; a < 5; ++a)
        for (size_t s : steps) {
            size_t sk[5] = {0, 0, 0, 0, 0}; sk[a] = s;
            printf("%9zu %9zu %9zu %9zu %9zu  %.3f\n", sk[0], sk[1], sk[2], sk[3], sk[4], run(sk)); fflush(stdout);
        }
    // random placements
    srand(12345);
    for (int k = 0; k < 40; ++k) {
        size_t sk[5];
        for (int a = 0; a < 5; ++a) sk[a] = ((size_t)rand() % (slack / 256)) * 256;
        printf("%9zu %9zu %9zu %9zu %9zu  %.3f\n", sk[0], sk[1], sk[2], sk[3], sk[4], run(sk)); fflush(stdout);
    }
    return 0;
}
