// Micro-benchmark: what does the y store of the plain-CSR SpMV cost, and what shapes it?  The traffic skeleton of
// tools/micro/mixstream.hip (values 16 B + indices 8 B per lane and slot, 4 slots in flight, dependent row pointers, x read once;
// one 512-row tile per workgroup, nontemporal stream loads) with the 4 KiB of y per tile written in different ways:
//   mode 0  not at all                      mode 1  every tile, 16 B per lane, nontemporal (what the kernel does)
//   mode 2  every 2nd tile   mode 3  every 4th tile   mode 4  every 8th tile             (is the cost linear in the bytes?)
//   mode 5  every tile, but the 16 pieces of 256 B go to addresses 64 KiB apart (one memory channel's worth per workgroup, if the
//           interleave is 256 B x 256 channels; the array is then a permutation of y -- mechanism probe, not a usable layout)
//   mode 6  as 5 with 4 KiB pieces 1 MiB apart... (whole tile contiguous, but consecutive WORKGROUPS write 1 MiB apart)
//   mode 7  every tile, plain (cacheable) stores          mode 8  every tile written TWICE (second time 2 KiB shifted: 2x the bytes)
//   mode 9  the y tile is written by the workgroup that runs 2048 tiles later (the store is far from the tile's own reads in address)
// build: hipcc --offload-arch=gfx950 -O3 tools/micro/writecost.hip -o tools/micro/writecost
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned int u4 __attribute__((ext_vector_type(4)));
typedef unsigned int u2 __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(256) void mix(const char* val, const char* col, const int* rp, const char* x, char* y, int vb, int mode, long long ntiles, unsigned* out) {
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
            v[u] = __builtin_nontemporal_load(reinterpret_cast<const u4*>(vbase + o));
            c[u] = __builtin_nontemporal_load(reinterpret_cast<const u2*>(cbase + o / 2));
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) acc ^= v[u].x ^ v[u].y ^ v[u].z ^ v[u].w ^ c[u].x ^ c[u].y;
    }
    const u4 xv = *reinterpret_cast<const u4*>(x + q * 4096 + t * 16); acc ^= xv.x ^ xv.w;
    u4 yv; yv.x = acc; yv.y = t; yv.z = 0; yv.w = 1;
    long long yo = q * 4096 + t * 16;
    bool doit = mode != 0;
    if (mode == 2) doit = (q & 1) == 0;
    if (mode == 3) doit = (q & 3) == 0;
    if (mode == 4) doit = (q & 7) == 0;
    if (mode == 5) {                                   // tile q's piece j (256 B) -> ((q / 16) * 16 + j) * 4096 + (q % 16) * 256
        const int j = t >> 4;
        yo = ((q >> 4) * 16 + j) * 4096 + (q & 15) * 256 + (t & 15) * 16;
    }
    if (mode == 6) {                                   // consecutive workgroups 1 MiB apart: q -> (q % 256) * 256 + (q / 256) % 256 within blocks of 65536 tiles
        const long long blk = q >> 16, r = q & 65535;
        yo = (blk * 65536 + (r & 255) * 256 + (r >> 8)) * 4096 + t * 16;
    }
    if (mode == 9) yo = ((q + 2048) % ntiles) * 4096 + t * 16;
    if (doit) {
        u4* yp = reinterpret_cast<u4*>(y + yo);
        if (mode == 7) *yp = yv; else __builtin_nontemporal_store(yv, yp);
        if (mode == 8) __builtin_nontemporal_store(yv, reinterpret_cast<u4*>(y + (ntiles + q) * 4096 + t * 16));
    }
    else if (acc == 0x12345678u) out[0] = acc;
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

int main(int argc, char** argv) {
    const long long ntiles = argc > 1 ? atoll(argv[1]) : 262144;          // 512^3
    const int vb = 28672;
    char *val, *col, *x, *y; int* rp; unsigned* out;
    CK(hipMalloc(&val, (size_t)ntiles * vb + 8192)); CK(hipMalloc(&col, (size_t)ntiles * vb / 2 + 8192));
    CK(hipMalloc(&rp, (size_t)ntiles * 2048 + 8192)); CK(hipMalloc(&x, (size_t)ntiles * 4096 + 8192)); CK(hipMalloc(&y, (size_t)ntiles * 8192 + 8192));
    CK(hipMalloc(&out, 64));
    CK(hipMemset(val, 1, (size_t)ntiles * vb + 8192)); CK(hipMemset(col, 2, (size_t)ntiles * vb / 2 + 8192));
    CK(hipMemset(rp, 0, (size_t)ntiles * 2048 + 8192)); CK(hipMemset(x, 3, (size_t)ntiles * 4096 + 8192));
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    printf("%lld tiles\nmode  ms (median of 7, three rounds)\n", ntiles);
    for (int round = 0; round < 3; ++round)
        for (int mode = 0; mode <= 9; ++mode) {
            std::vector<float> ms;
            for (int r = 0; r < 8; ++r) {
                CK(hipEventRecord(e0, 0));
                hipLaunchKernelGGL(mix, dim3((unsigned)ntiles), dim3(256), 0, 0, val, col, rp, x, y, vb, mode, ntiles, out);
                CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
                float m; CK(hipEventElapsedTime(&m, e0, e1));
                if (r) ms.push_back(m);
            }
            std::sort(ms.begin(), ms.end());
            printf("%d  %.3f\n", mode, ms[3]); fflush(stdout);
        }
    return 0;
}
