// Micro-benchmark: the solving wave's k-neighbour exchange (lane l takes the value of lane l - 8) without the LDS crossbar:
// ds_bpermute answers in ~100+ cycles and, because a wave's LDS operations return in order, also waits for the ring write and the
// counter store of the step before.  gfx950 has v_permlane16_swap / v_permlane32_swap (rows of 16 lanes, halves of 32): lane l - 8 is
// a DPP row shift for the upper half of every row of 16 and, for the lower half, the upper half of the ROW BEFORE -- a row shift built
// from the two swaps and selects, all VALU.  Checks the network against ds_bpermute, then times the dependent chain of the
// step (exchange + selects + three mul / subtract) with either exchange, with and without the step's LDS writes.
// build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/micro/shr8.hip -o tools/micro/shr8
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned v2u __attribute__((ext_vector_type(2)));
__device__ __forceinline__ double bperm(int idx, double v) {
    int lo = __builtin_amdgcn_ds_bpermute(idx, __double2loint(v)), hi = __builtin_amdgcn_ds_bpermute(idx, __double2hiint(v));
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double shr1(double v) {
    int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x111, 0xf, 0xf, false), hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x111, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
// one dword: out[l] = in[l - 8] for l >= 8 (lanes 0-7: unspecified)
__device__ __forceinline__ unsigned shr8_u32(unsigned x, bool row3, bool upper) {
    const unsigned U = (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xf, false);      // row_shr:8 -- lanes 8-15 of a row <- its lanes 0-7
    const unsigned D = (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x108, 0xf, 0xf, false);      // row_shl:8 -- lanes 0-7 of a row <- its lanes 8-15
    const v2u s16 = __builtin_amdgcn_permlane16_swap(D, D, false, false);      // {[D0 D0 D2 D2], [D1 D1 D3 D3]} by rows
    const v2u s32 = __builtin_amdgcn_permlane32_swap(s16.x, s16.y, false, false);   // {[D0 D0 D1 D1], [D2 D2 D3 D3]}
    const unsigned E = row3 ? s16.x : s32.x;                                        // row r <- D of row r - 1
    return upper ? U : E;
}
__device__ __forceinline__ double shr8(double v, bool row3, bool upper) {
    return __hiloint2double((int)shr8_u32((unsigned)__double2hiint(v), row3, upper), (int)shr8_u32((unsigned)__double2loint(v), row3, upper));
}
__global__ void check(int* bad, unsigned* dump) {
    const int l = threadIdx.x;
    const double v = 1000.0 + l;
    const double a = bperm(max(l - 8, 0) * 4, v), b = shr8(v, (l >> 4) == 3, (l & 8) != 0);
    if (l >= 8 && a != b) atomicAdd(bad, 1);
    const v2u s16 = __builtin_amdgcn_permlane16_swap((unsigned)l, (unsigned)(l + 100), false, false);
    const v2u s32 = __builtin_amdgcn_permlane32_swap((unsigned)l, (unsigned)(l + 100), false, false);
    dump[l] = s16.x; dump[64 + l] = s16.y; dump[128 + l] = s32.x; dump[192 + l] = s32.y; dump[256 + l] = (unsigned)b;
}
template <int MODE>
__global__ __launch_bounds__(256) void k(int steps, long long* out, double* sink) {
    __shared__ double ring[4 * 32 * 64];
    __shared__ int prog[4];
    const int l = threadIdx.x & 63, q = threadIdx.x >> 6, jl = l & 7, kl = l >> 3;
    for (int i = threadIdx.x; i < 4 * 32 * 64; i += 256) ring[i] = 0.001 * i;
    __syncthreads();
    double* mine = ring + q * 32 * 64 + l;
    const int idx8 = max(l - 8, 0) * 4;
    const bool row3 = (l >> 4) == 3, upper = (l & 8) != 0;
    double y = 1.0 + l;
    double a1 = 0.25, a2 = 0.125, a3 = 0.0625;
    if (l == 63) a3 = 0.0;
    const long long w0 = wall_clock64();
    for (int t = 0; t < steps; t += 8) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int tt = t + u;
            double yj = shr1(y), yk = (MODE & 1) ? shr8(y, row3, upper) : bperm(idx8, y);
            if (jl == 0) yj = 0.5;
            if (kl == 0) yk = 0.25;
            yj = a2 != 0.0 ? yj : 0.0; yk = a3 != 0.0 ? yk : 0.0;
            const double yi = a1 != 0.0 ? y : 0.0;
            double s = 3.0;
            s = s - a3 * yk; s = s - a2 * yj; s = s - a1 * yi;
            y = s;
            if (MODE & 2) {
                mine[(tt & 31) * 64] = s;
                asm volatile("" ::: "memory"); __hip_atomic_store(&prog[q], tt + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); asm volatile("" ::: "memory");
            }
        }
    }
    const long long w1 = wall_clock64();
    if (l == 0 && q == 0 && blockIdx.x == 0) out[0] = w1 - w0;
    if (y == 12345.678) sink[0] = y;
}
template <int MODE> void run(const char* name, long long* out, double* sink, int blocks) {
    const int steps = 1 << 15;
    for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(256), 0, 0, steps, out, sink); (void)hipDeviceSynchronize(); }
    long long h; (void)hipMemcpy(&h, out, 8, hipMemcpyDeviceToHost);
    printf("%-64s blocks %3d: %6.1f ns per step\n", name, blocks, h * 10.0 / steps); fflush(stdout);
}
int main() {
    long long* out; double* sink; int* bad; unsigned* dump;
    (void)hipMalloc(&out, 8); (void)hipMalloc(&sink, 8); (void)hipMalloc(&bad, 4); (void)hipMalloc(&dump, 320 * 4);
    (void)hipMemset(bad, 0, 4);
    hipLaunchKernelGGL(check, dim3(1), dim3(64), 0, 0, bad, dump);
    int hb; unsigned hd[320]; (void)hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost); (void)hipMemcpy(hd, dump, sizeof hd, hipMemcpyDeviceToHost);
    printf("shift-by-8 network vs ds_bpermute: %d lanes differ\n", hb);
    const char* names[5] = {"permlane16_swap(l, l+100).x", "permlane16_swap.y", "permlane32_swap.x", "permlane32_swap.y", "shr8(1000 + l)"};
    for (int r = 0; r < 5; ++r) { printf("%-28s", names[r]); for (int l = 0; l < 64; l += 4) printf(" %4u", hd[64 * r + l]); printf("\n"); }
    for (int blocks : {1, 256}) {
        run<0>("chain with ds_bpermute", out, sink, blocks);
        run<1>("chain with the VALU shift-by-8", out, sink, blocks);
        run<2>("chain with ds_bpermute + ring write + counter (all lanes)", out, sink, blocks);
        run<3>("chain with the VALU shift-by-8 + ring write + counter (all lanes)", out, sink, blocks);
    }
    return 0;
}
