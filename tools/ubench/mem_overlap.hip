// Micro-benchmark (dev tool): issue cost of 1-KiB wave loads and LDS reads beside v_mfma_f32_16x16x4_f32.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0)

template <int MODE, int NL>
__global__ __launch_bounds__(256) void k(float *out, unsigned long long *cyc, const f4 *src, int iters, float x)
{
    __shared__ f4 lds[2048];
    for (int i = threadIdx.x; i < 2048; i += 256) lds[i] = (f4){x, x, x, x};
    __syncthreads();
    float a[8], b[8];
    for (int i = 0; i < 8; i++) { a[i] = x + i + threadIdx.x; b[i] = x * i - threadIdx.x; }
    f4 acc[8];
    for (int i = 0; i < 8; i++) acc[i] = (f4){x, x, x, x};
    f4 ld[8];
    for (int i = 0; i < 8; i++) ld[i] = (f4){0, 0, 0, 0};
    const int lane = threadIdx.x & 63;
    const f4 *base = src + (threadIdx.x >> 6) * 64 * 32;
    __builtin_amdgcn_sched_barrier(0);
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_sched_barrier(0);
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            acc[i] = MFMA(a[i] + ld[i][0], b[i], acc[i]);   // consumes the load of the PREVIOUS iteration
            if (MODE == 1 && i < NL) ld[i] = base[((it & 3) * 8 + i) * 64 + lane];
            if (MODE == 2 && i < NL) ld[i] = lds[((it & 3) * 8 + i) * 64 + lane];
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    f4 r = {0, 0, 0, 0};
    for (int i = 0; i < 8; i++) r += acc[i] + ld[i];
    out[blockIdx.x * 256 + threadIdx.x] = r[0] + r[1] + r[2] + r[3];
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int MODE, int NL>
void run(const char *name, const f4 *src)
{
    const int grid = 256, iters = 200;
    float *out; unsigned long long *cyc;
    (void)hipMalloc(&out, grid * 256 * 4); (void)hipMalloc(&cyc, grid * 4 * 8);
    for (int rep = 0; rep < 3; rep++) k<MODE, NL><<<grid, 256>>>(out, cyc, src, iters, 1.0f);
    (void)hipDeviceSynchronize();
    std::vector<unsigned long long> h(grid * 4);
    (void)hipMemcpy(h.data(), cyc, grid * 4 * 8, hipMemcpyDeviceToHost);
    double s = 0; for (auto v : h) s += (double)v;
    printf("%-50s %7.2f cycles per 8 MFMAs (%d loads) -> %.2f per load\n", name, s / h.size() / iters, NL,
           NL ? (s / h.size() / iters - 8 * 35.0) / NL : 0.0);
    (void)hipFree(out); (void)hipFree(cyc);
}

int main()
{
    f4 *src; (void)hipMalloc(&src, 4 * 64 * 32 * 16); (void)hipMemset(src, 0, 4 * 64 * 32 * 16);
    run<0, 0>("8 MFMAs (+v_add feeding them)", src);
    run<1, 1>("8 MFMAs + 1 global_load_dwordx4", src); run<1, 2>("8 MFMAs + 2 global_load_dwordx4", src);
    run<1, 4>("8 MFMAs + 4 global_load_dwordx4", src); run<1, 8>("8 MFMAs + 8 global_load_dwordx4", src);
    run<2, 1>("8 MFMAs + 1 ds_read_b128", src); run<2, 2>("8 MFMAs + 2 ds_read_b128", src);
    run<2, 4>("8 MFMAs + 4 ds_read_b128", src); run<2, 8>("8 MFMAs + 8 ds_read_b128", src);
    return 0;
}
