// Micro-benchmark (dev tool) behind profiles/r02_ab_runs.txt items 6 and 9: one hidden tile's worth of layer-1 work per
// iteration -- 11 one-KiB weight fragments streamed from global memory, the input operand from LDS -- in three forms:
//   16x16x4, 64-bit     v_mfma_f32_16x16x4_f32, two 16-frame tiles (84 MFMAs), weight loads with 64-bit per-lane addresses
//                       (`global_load v, v[addr64], off`: what hipcc chose for fragments beyond the 13-bit immediate range)
//   16x16x4, scalar     the same with every group of four fragments on its own SGPR base (`global_load v, v_lane, s[base] offset:imm`)
//   32x32x2, scalar     v_mfma_f32_32x32x2_f32 on one 32-frame tile: the same FLOPs as 42 MFMAs of twice the length; the same
//                       11 weight fragments, HALF the LDS operand reads
// One wave per SIMD (256 threads, one workgroup per CU), s_memtime around the loop.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(1))) const f4 gf4;

__device__ __forceinline__ gf4 *scalar_ptr(const f4 *p)
{
    const unsigned long long u = reinterpret_cast<unsigned long long>(p);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u), hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
    return (gf4 *)(((unsigned long long)hi << 32) | lo);
}

// MODE 0: 16x16x4 + 64-bit addresses, 1: 16x16x4 + scalar bases, 2: 32x32x2 + scalar bases.  LDSB: B operand from LDS.
template <int MODE, bool LDSB>
__global__ __launch_bounds__(256) void k(float *out, unsigned long long *cyc, const f4 *src, int iters, float x)
{
    __shared__ f4 xb[2 * 11 * 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 2 * 11 * 64; i += 256) xb[i] = (f4){x + i, x, x - i, x};
    __syncthreads();
    f4 acc0 = {x, x, x, x}, acc1 = {x, x, x, x};
    f16v big;
    for (int i = 0; i < 16; i++) big[i] = x + i;
    f4 cur[11], nxt[11];
    for (int i = 0; i < 11; i++) cur[i] = nxt[i] = (f4){x + i, x, x, x};
    float breg[8];
    for (int i = 0; i < 8; i++) breg[i] = x * i - lane;
    const f4 *wbase = src + wave * 64 * 16 * 4;              // per-wave region, 4 tiles of 16 fragments
    __builtin_amdgcn_sched_barrier(0);
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_sched_barrier(0);
    for (int it = 0; it < iters; it++) {
        const f4 *t = wbase + (it & 3) * 64 * 16;
        // the LDS operand of group g + 1 is requested before the MFMAs of group g, as in the kernel's hidden loop
        f4 bq[2][2];
        if (LDSB) {
            bq[0][0] = xb[lane];
            if (MODE != 2) bq[0][1] = xb[11 * 64 + lane];
        }
#pragma unroll
        for (int g = 0; g < 11; g++) {
            if (MODE == 0) nxt[g] = t[g * 64 + lane];                                   // per-lane pointer arithmetic
            else nxt[g] = scalar_ptr(t + (g & ~3) * 64)[(g & 3) * 64 + lane];           // scalar base per four fragments
            f4 b0, b1;
            if (LDSB) {
                if (g + 1 < 11) {
                    bq[(g + 1) & 1][0] = xb[(g + 1) * 64 + lane];
                    if (MODE != 2) bq[(g + 1) & 1][1] = xb[(11 + g + 1) * 64 + lane];
                }
                b0 = bq[g & 1][0];
                b1 = bq[g & 1][1];
            } else {
                b0 = (f4){breg[0], breg[1], breg[2], breg[3]};
                b1 = (f4){breg[4], breg[5], breg[6], breg[7]};
            }
#pragma unroll
            for (int j = 0; j < 4; j++) {
                if (4 * g + j < 42) {
                    if (MODE == 2) {
                        big = __builtin_amdgcn_mfma_f32_32x32x2f32(cur[g][j], b0[j], big, 0, 0, 0);
                    } else {
                        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(cur[g][j], b0[j], acc0, 0, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(cur[g][j], b1[j], acc1, 0, 0, 0);
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        asm volatile("" ::: "memory");
#pragma unroll
        for (int i = 0; i < 11; i++) cur[i] = nxt[i];
    }
    __builtin_amdgcn_sched_barrier(0);
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float r = acc0[0] + acc0[1] + acc1[2] + acc1[3];
    for (int i = 0; i < 16; i++) r += big[i];
    out[blockIdx.x * 256 + threadIdx.x] = r;
    if (lane == 0) cyc[blockIdx.x * 4 + wave] = t1 - t0;
}

template <int MODE, bool LDSB>
double run(const char *name, const f4 *src)
{
    const int grid = 256, iters = 200;
    float *out; unsigned long long *cyc;
    (void)hipMalloc(&out, grid * 256 * 4); (void)hipMalloc(&cyc, grid * 4 * 8);
    for (int rep = 0; rep < 3; rep++) k<MODE, LDSB><<<grid, 256>>>(out, cyc, src, iters, 1.0f);
    (void)hipDeviceSynchronize();
    std::vector<unsigned long long> h(grid * 4);
    (void)hipMemcpy(h.data(), cyc, grid * 4 * 8, hipMemcpyDeviceToHost);
    double s = 0; for (auto v : h) s += (double)v;
    const double per = s / h.size() / iters;
    // 84 MFMAs 16x16x4 (or 42 of 32x32x2) issue in 84 x 32 = 2688 cycles
    printf("%-58s %8.1f cycles per tile  (MFMA issue alone: 2688; overhead %+6.1f)\n", name, per, per - 2688.0);
    (void)hipFree(out); (void)hipFree(cyc);
    return per;
}

int main()
{
    f4 *src; (void)hipMalloc(&src, 4 * 4 * 64 * 16 * 16); (void)hipMemset(src, 0, 4 * 4 * 64 * 16 * 16);
    run<0, false>("16x16x4 x2 tiles, 64-bit load addresses, B in registers", src);
    run<1, false>("16x16x4 x2 tiles, scalar-base loads,     B in registers", src);
    run<2, false>("32x32x2 x1 tile,  scalar-base loads,     B in registers", src);
    run<0, true>("16x16x4 x2 tiles, 64-bit load addresses, B from LDS (22 reads)", src);
    run<1, true>("16x16x4 x2 tiles, scalar-base loads,     B from LDS (22 reads)", src);
    run<2, true>("32x32x2 x1 tile,  scalar-base loads,     B from LDS (11 reads)", src);
    return 0;
}
