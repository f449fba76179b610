// Micro-benchmark: the plain-CSR SpMV's memory traffic without its arithmetic -- per 512-row "tile" a workgroup reads VB bytes of
// a value stream (16 B per lane), VB/2 bytes of an index stream (8 B per lane), optionally 2 KB of row pointers FIRST (dependent:
// the stream addresses wait for them), optionally 4 KB of x (16 B per lane, plain loads) and optionally writes 4 KB of y
// (nontemporal).  One tile per workgroup, workgroups in index order; each of the 4 waves owns a contiguous quarter of the tile's
// streams and keeps U pair-loads in flight, as spmv_wave_kernel does.  `mis`: byte offset added to the stream bases (windows that
// do not start on a memory line).  Prints GB/s of ALL bytes moved.
// build: hipcc --offload-arch=gfx950 -O3 tools/micro/mixstream.hip -o tools/micro/mixstream
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned int u4 __attribute__((ext_vector_type(4)));
typedef unsigned int u2 __attribute__((ext_vector_type(2)));

template <int U, bool NT>
__global__ __launch_bounds__(256) void mix(const char* val, const char* col, const int* rp, const char* x, char* y, int vb, int flags, int mis, unsigned* out) {
    const int t = threadIdx.x, l = t & 63, w = t >> 6;
    const long long q = blockIdx.x;
    unsigned acc = 0;
    int shift = 0;
    if (flags & 1) {                                          // dependent row pointers: the stream offset comes out of memory
        const int p = rp[q * 512 + 2 * t];
        const int k0 = __builtin_amdgcn_readfirstlane(p);
        shift = k0 & 16; acc ^= p;                            // (always 0 or 16: keeps the addresses in range, the compiler cannot know)
    }
    const int wv = vb / 4;                                    // bytes of the value stream per wave
    const char* vbase = val + q * vb + (long long)w * wv + mis + shift;
    const char* cbase = col + q * (vb / 2) + (long long)w * (wv / 2) + mis / 2 + shift / 2;
    for (int off = 0; off < wv; off += U * 1024) {
        u4 v[U]; u2 c[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int o = min(off + u * 1024 + l * 16, wv - 16);
            const u4* pv = reinterpret_cast<const u4*>(vbase + o);
            const u2* pc = reinterpret_cast<const u2*>(cbase + o / 2);
            v[u] = NT ? __builtin_nontemporal_load(pv) : *pv;
            c[u] = NT ? __builtin_nontemporal_load(pc) : *pc;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) acc ^= v[u].x ^ v[u].y ^ v[u].z ^ v[u].w ^ c[u].x ^ c[u].y;
    }
    if (flags & 2) { const u4 xv = *reinterpret_cast<const u4*>(x + q * 4096 + t * 16); acc ^= xv.x ^ xv.w; }
    if (flags & 4) {
        u4 yv; yv.x = acc; yv.y = t; yv.z = 0; yv.w = 1;
        u4* yp = reinterpret_cast<u4*>(y + q * 4096 + t * 16);
        const int fl = flags >> 4;                             // store flavour: 0 nt, 1 plain, 2 sc1, 3 sc0 sc1, 4 nt sc1, 5 sc0
        if (fl == 0) __builtin_nontemporal_store(yv, yp);
        else if (fl == 1) *yp = yv;
        else if (fl == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1" : : "v"(yp), "v"(yv) : "memory");
        else if (fl == 3) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" : : "v"(yp), "v"(yv) : "memory");
        else if (fl == 4) asm volatile("global_store_dwordx4 %0, %1, off sc1 nt" : : "v"(yp), "v"(yv) : "memory");
        else asm volatile("global_store_dwordx4 %0, %1, off sc0" : : "v"(yp), "v"(yv) : "memory");
    }
    else if (acc == 0x12345678u) out[0] = acc;
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

int main(int argc, char** argv) {
    const long long ntiles = argc > 1 ? atoll(argv[1]) : 262144;          // 512^3: 262144 tiles
    const int vb = 28672;                                                 // 3584 entries x 8 B
    char *val, *col, *x, *y; int* rp; unsigned* out;
    CK(hipMalloc(&val, (size_t)ntiles * vb + 8192)); CK(hipMalloc(&col, (size_t)ntiles * vb / 2 + 8192));
    CK(hipMalloc(&rp, (size_t)ntiles * 2048 + 8192)); CK(hipMalloc(&x, (size_t)ntiles * 4096 + 8192)); CK(hipMalloc(&y, (size_t)ntiles * 4096 + 8192));
    CK(hipMalloc(&out, 64));
    CK(hipMemset(val, 1, (size_t)ntiles * vb + 8192)); CK(hipMemset(col, 2, (size_t)ntiles * vb / 2 + 8192));
    CK(hipMemset(rp, 0, (size_t)ntiles * 2048 + 8192)); CK(hipMemset(x, 3, (size_t)ntiles * 4096 + 8192));
    CK(hipDeviceSynchronize());
    printf("%lld tiles; flags: 1 dependent row pointers, 2 x read, 4 y write\n U nt flags mis   ms     GB/s(all bytes)\n", ntiles);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int mis : {48})
        for (int flags : {0, 4, 4 + 16, 4 + 32, 4 + 48, 4 + 64, 4 + 80, 7, 7 + 16, 7 + 32, 7 + 48, 7 + 64, 7 + 80})
            for (int cfg = 0; cfg < 4; ++cfg) {
                const int U = cfg < 2 ? 4 : 7; const bool nt = cfg & 1;
                std::vector<float> ms;
                for (int r = 0; r < 6; ++r) {
                    CK(hipEventRecord(e0, 0));
                    if (U == 4 && !nt) hipLaunchKernelGGL((mix<4, false>), dim3((unsigned)ntiles), dim3(256), 0, 0, val, col, rp, x, y, vb, flags, mis, out);
                    if (U == 4 && nt) hipLaunchKernelGGL((mix<4, true>), dim3((unsigned)ntiles), dim3(256), 0, 0, val, col, rp, x, y, vb, flags, mis, out);
                    if (U == 7 && !nt) hipLaunchKernelGGL((mix<7, false>), dim3((unsigned)ntiles), dim3(256), 0, 0, val, col, rp, x, y, vb, flags, mis, out);
                    if (U == 7 && nt) hipLaunchKernelGGL((mix<7, true>), dim3((unsigned)ntiles), dim3(256), 0, 0, val, col, rp, x, y, vb, flags, mis, out);
                    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
                    float m; CK(hipEventElapsedTime(&m, e0, e1));
                    if (r) ms.push_back(m);
                }
                std::sort(ms.begin(), ms.end());
                const double bytes = (double)ntiles * (vb * 1.5 + ((flags & 1) ? 2048 : 0) + ((flags & 2) ? 4096 : 0) + ((flags & 4) ? 4096 : 0));
                printf("%2d %2d %5d(st %d) %3d  %6.3f  %7.1f\n", U, (int)nt, flags & 15, flags >> 4, mis, ms[2], bytes / (ms[2] * 1e-3) / 1e9); fflush(stdout);
            }
    return 0;
}
