// Micro-benchmark (dev tool): cycles per v_mfma_f32_16x16x4_f32 for the operand/accumulator
// patterns the fused kernel uses.  One wave per SIMD (256 threads, 1 block per CU).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0)

template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, unsigned long long *cyc, int iters, float x)
{
    float a[12], b[24];
    for (int i = 0; i < 12; i++) a[i] = x + i + threadIdx.x;
    for (int i = 0; i < 24; i++) b[i] = x * i - threadIdx.x;
    f4 acc[18];
    for (int i = 0; i < 18; i++) acc[i] = (f4){x, x, x, x};
    __builtin_amdgcn_sched_barrier(0);
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_sched_barrier(0);
    for (int it = 0; it < iters; it++) {
        if (MODE == 0) {            // 72 MFMAs, 18 independent accumulators (layer-2 pattern)
#pragma unroll
            for (int r = 0; r < 4; r++)
#pragma unroll
                for (int i = 0; i < 18; i++) acc[i] = MFMA(a[(i >> 1) % 12], b[r * 2 + (i & 1)], acc[i]);
        } else if (MODE == 1) {     // 84 MFMAs, 2 alternating accumulators (layer-1 pattern)
#pragma unroll
            for (int s = 0; s < 42; s++) {
                acc[0] = MFMA(a[s % 12], b[s % 24], acc[0]);
                acc[1] = MFMA(a[s % 12], b[(s + 7) % 24], acc[1]);
            }
        } else if (MODE == 2) {     // 84 MFMAs, 4 accumulators
#pragma unroll
            for (int s = 0; s < 21; s++) {
                acc[0] = MFMA(a[s % 12], b[s % 24], acc[0]);
                acc[1] = MFMA(a[s % 12], b[(s + 7) % 24], acc[1]);
                acc[2] = MFMA(a[(s + 1) % 12], b[s % 24], acc[2]);
                acc[3] = MFMA(a[(s + 1) % 12], b[(s + 7) % 24], acc[3]);
            }
        } else if (MODE == 3) {     // 84 MFMAs, a single dependent chain
#pragma unroll
            for (int s = 0; s < 84; s++) acc[0] = MFMA(a[s % 12], b[s % 24], acc[0]);
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    f4 r = {0, 0, 0, 0};
    for (int i = 0; i < 18; i++) r += acc[i];
    out[blockIdx.x * 256 + threadIdx.x] = r[0] + r[1] + r[2] + r[3];
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int MODE>
void run(const char *name, int per_iter)
{
    const int grid = 256, iters = 200;
    float *out; unsigned long long *cyc;
    hipMalloc(&out, grid * 256 * 4); hipMalloc(&cyc, grid * 4 * 8);
    for (int rep = 0; rep < 3; rep++) k<MODE><<<grid, 256>>>(out, cyc, iters, 1.0f);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(grid * 4);
    hipMemcpy(h.data(), cyc, grid * 4 * 8, hipMemcpyDeviceToHost);
    double s = 0; for (auto v : h) s += (double)v;
    printf("%-48s %7.2f cycles/MFMA\n", name, s / h.size() / iters / per_iter);
    hipFree(out); hipFree(cyc);
}

int main()
{
    run<0>("72 MFMAs, 18 independent accumulators", 72);
    run<1>("84 MFMAs, 2 alternating accumulators", 84);
    run<2>("84 MFMAs, 4 accumulators", 84);
    run<3>("84 MFMAs, 1 dependent chain", 84);
    return 0;
}
