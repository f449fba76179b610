// Micro-benchmark: what read rate does this MI355X sustain on a once-read stream far beyond the 256 MiB Infinity Cache, as a
// function of HOW the stream is cut over workgroups?  (The plain-CSR SpMV is 92 % such a stream: 12 bytes per entry.)
//   map 0  one tile per workgroup, workgroups dispatched in index order (grid = ntiles)        -- spmv_wave_kernel's shape
//   map 1  persistent grid, G workgroups per CU, tile q = b, b + grid, ...                      -- the BLAS-1 kernels' shape
//   map 2  runs of R consecutive tiles per workgroup, runs in index order                      -- spmv_pipe_kernel's shape
// A tile is TB bytes; a lane issues U loads of W bytes (W = 16 or 8), U * 256 * W bytes in flight per workgroup, then the next
// batch of the tile.  nt: nontemporal loads.  Prints GB/s per configuration (median of 5 timed launches).
// build: hipcc --offload-arch=gfx950 -O3 tools/micro/readstream.hip -o tools/micro/readstream
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned int u4 __attribute__((ext_vector_type(4)));
typedef unsigned int u2 __attribute__((ext_vector_type(2)));

template <int W, int U, bool NT>
__device__ __forceinline__ unsigned tile_sum(const char* base, int tb, int t) {
    unsigned acc = 0;
    for (int off = 0; off < tb; off += U * 256 * W) {
        if constexpr (W == 16) {
            u4 v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int o = min(off + u * 256 * W + t * W, tb - W);
                const u4* p = reinterpret_cast<const u4*>(base + o);
                v[u] = NT ? __builtin_nontemporal_load(p) : *p;
            }
#pragma unroll
            for (int u = 0; u < U; ++u) acc ^= v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
        } else {
            u2 v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int o = min(off + u * 256 * W + t * W, tb - W);
                const u2* p = reinterpret_cast<const u2*>(base + o);
                v[u] = NT ? __builtin_nontemporal_load(p) : *p;
            }
#pragma unroll
            for (int u = 0; u < U; ++u) acc ^= v[u].x ^ v[u].y;
        }
    }
    return acc;
}

template <int W, int U, bool NT>
__global__ __launch_bounds__(256) void rd(const char* buf, long long ntiles, int tb, int map, int run, unsigned* out) {
    const int t = threadIdx.x;
    unsigned acc = 0;
    if (map == 0) {
        acc = tile_sum<W, U, NT>(buf + (long long)blockIdx.x * tb, tb, t);
    } else if (map == 1) {
        for (long long q = blockIdx.x; q < ntiles; q += gridDim.x) acc ^= tile_sum<W, U, NT>(buf + q * tb, tb, t);
    } else {
        for (long long q = (long long)blockIdx.x * run; q < std::min<long long>(ntiles, ((long long)blockIdx.x + 1) * run); ++q)
            acc ^= tile_sum<W, U, NT>(buf + q * tb, tb, t);
    }
    if (acc == 0x12345678u) out[0] = acc;
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int W, int U, bool NT>
static double run_one(const char* buf, long long ntiles, int tb, int map, int par, unsigned* out) {
    dim3 grid;
    int run = 1;
    if (map == 0) grid = dim3((unsigned)ntiles);
    else if (map == 1) grid = dim3((unsigned)std::min<long long>(ntiles, 256ll * par));
    else { run = par; grid = dim3((unsigned)((ntiles + run - 1) / run)); }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    std::vector<float> ms;
    for (int r = 0; r < 6; ++r) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL((rd<W, U, NT>), grid, dim3(256), 0, 0, buf, ntiles, tb, map, run, out);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float m; hipEventElapsedTime(&m, e0, e1);
        if (r) ms.push_back(m);
    }
    std::sort(ms.begin(), ms.end());
    hipEventDestroy(e0); hipEventDestroy(e1);
    return (double)ntiles * tb / (ms[ms.size() / 2] * 1e-3) / 1e9;
}

int main(int argc, char** argv) {
    const double gb = argc > 1 ? atof(argv[1]) : 12.0;
    const int tb = argc > 2 ? atoi(argv[2]) : 43008;          // col + val bytes of a 512-row tile of the 7-point operator
    const long long ntiles = (long long)(gb * 1e9 / tb);
    char* buf; unsigned* out;
    CK(hipMalloc(&buf, (size_t)ntiles * tb + 4096)); CK(hipMalloc(&out, 64));
    CK(hipMemset(buf, 1, (size_t)ntiles * tb + 4096));
    CK(hipDeviceSynchronize());
    printf("read stream of %.2f GB in tiles of %d bytes (%lld tiles)\n", (double)ntiles * tb / 1e9, tb, ntiles);
    printf("map par   W  U nt   GB/s\n");
#define ROW(W_, U_, NT_, map, par) printf("%3d %3d  %2d %2d %2d  %7.1f\n", map, par, W_, U_, (int)NT_, run_one<W_, U_, NT_>(buf, ntiles, tb, map, par, out)); fflush(stdout)
    for (int map_par : {0, 102, 104, 106, 108, 116, 202, 208}) {
        const int map = map_par / 100, par = map_par % 100;
        ROW(16, 4, false, map, par); ROW(16, 4, true, map, par);
        ROW(16, 8, false, map, par); ROW(16, 8, true, map, par);
        ROW(8, 8, false, map, par); ROW(8, 8, true, map, par);
        ROW(16, 2, false, map, par); ROW(16, 2, true, map, par);
    }
    hipFree(buf); hipFree(out);
    return 0;
}
