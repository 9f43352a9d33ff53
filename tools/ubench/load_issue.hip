// Micro-benchmark (dev tool): what does a 1-KiB wave load COST the issuing wave when its data is not
// needed for ~2700 cycles (one 84-MFMA phase later)?  spread = one load behind every 8-MFMA group,
// clump = all loads first.  One wave per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0)

template <int MODE, int NL>
__global__ __launch_bounds__(256) void k(float *out, unsigned long long *cyc, const f4 *src, int iters, float x)
{
    float b[8];
    for (int i = 0; i < 8; i++) b[i] = x * i - threadIdx.x;
    f4 acc0 = {x, x, x, x}, acc1 = {x, x, x, x};
    f4 cur[12], nxt[12];
    for (int i = 0; i < 12; i++) cur[i] = nxt[i] = (f4){x + i, x, x, x};
    const int lane = threadIdx.x & 63;
    const f4 *base = src + (threadIdx.x >> 6) * 64 * 16 * 4;
    __builtin_amdgcn_sched_barrier(0);
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_sched_barrier(0);
    for (int it = 0; it < iters; it++) {
        const f4 *t = base + (it & 3) * 64 * 16;
        if (MODE == 2) {
#pragma unroll
            for (int i = 0; i < NL; i++) nxt[i] = t[i * 64 + lane];
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int g = 0; g < 11; g++) {
            if (MODE == 1 && g < NL) nxt[g] = t[g * 64 + lane];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                if (4 * g + j < 42) {
                    acc0 = MFMA(cur[g][j], b[j], acc0);
                    acc1 = MFMA(cur[g][j], b[j + 4], acc1);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        asm volatile("" ::: "memory");
#pragma unroll
        for (int i = 0; i < 12; i++) cur[i] = (MODE == 0 || i >= NL) ? cur[i] : nxt[i];
    }
    __builtin_amdgcn_sched_barrier(0);
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    f4 r = acc0 + acc1;
    out[blockIdx.x * 256 + threadIdx.x] = r[0] + r[1] + r[2] + r[3];
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int MODE, int NL>
double run(const char *name, const f4 *src, double base)
{
    const int grid = 256, iters = 200;
    float *out; unsigned long long *cyc;
    (void)hipMalloc(&out, grid * 256 * 4); (void)hipMalloc(&cyc, grid * 4 * 8);
    for (int rep = 0; rep < 3; rep++) k<MODE, NL><<<grid, 256>>>(out, cyc, src, iters, 1.0f);
    (void)hipDeviceSynchronize();
    std::vector<unsigned long long> h(grid * 4);
    (void)hipMemcpy(h.data(), cyc, grid * 4 * 8, hipMemcpyDeviceToHost);
    double s = 0; for (auto v : h) s += (double)v;
    double per = s / h.size() / iters;
    printf("%-44s %8.1f cycles per 84 MFMAs", name, per);
    if (NL) printf("  (+%.1f per load)", (per - base) / NL);
    printf("\n");
    (void)hipFree(out); (void)hipFree(cyc);
    return per;
}

int main()
{
    f4 *src; (void)hipMalloc(&src, 4 * 4 * 64 * 16 * 16); (void)hipMemset(src, 0, 4 * 4 * 64 * 16 * 16);
    double b = run<0, 0>("84 MFMAs, no loads", src, 0);
    run<1, 4>("spread, 4 loads", src, b); run<1, 8>("spread, 8 loads", src, b); run<1, 11>("spread, 11 loads", src, b);
    run<2, 4>("clump, 4 loads", src, b); run<2, 8>("clump, 8 loads", src, b); run<2, 11>("clump, 11 loads", src, b);
    return 0;
}
