// Micro-benchmark (dev tool): does VALU work overlap with v_mfma_f32_16x16x4_f32 issued by the SAME wave
// (one wave per SIMD), and what do the sigmoid's instructions cost?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0)

template <int MODE, int NV>
__global__ __launch_bounds__(256) void k(float *out, unsigned long long *cyc, int iters, float x)
{
    float a[8], b[8];
    for (int i = 0; i < 8; i++) { a[i] = x + i + threadIdx.x; b[i] = x * i - threadIdx.x; }
    f4 acc[8];
    for (int i = 0; i < 8; i++) acc[i] = (f4){x, x, x, x};
    float v[8]; double d[8];
    for (int i = 0; i < 8; i++) { v[i] = x * (i + 1) + threadIdx.x * 1e-3f; d[i] = v[i]; }
    __builtin_amdgcn_sched_barrier(0);
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_sched_barrier(0);
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            if (MODE != 9) acc[i] = MFMA(a[i], b[i], acc[i]);
#pragma unroll
            for (int j = 0; j < NV; j++) {
                const int q = (i * NV + j) & 7;
                if (MODE == 1) v[q] = __builtin_fmaf(v[q], 1.0001f, 0.5f);
                if (MODE == 2) d[q] = __builtin_fma(d[q], 1.0001, 0.5);
                if (MODE == 3) d[q] = __builtin_amdgcn_rcp(d[q]);
                if (MODE == 4) d[q] = (double)v[q] + d[q];                 // cvt_f64_f32 + add_f64
                if (MODE == 5) v[q] += (float)d[q];                        // cvt_f32_f64 + add_f32
                if (MODE == 6) v[q] += (float)__double2int_rz(d[q]);       // cvt_i32_f64 + cvt_f32_i32 + add
                if (MODE == 7) d[q] = __builtin_amdgcn_div_fixup(d[q], 3.0, 1.0);
                if (MODE == 8) v[q] = __builtin_amdgcn_rcpf(v[q]);
                if (MODE == 9) v[q] = __builtin_fmaf(v[q], 1.0001f, 0.5f);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    f4 r = {0, 0, 0, 0};
    for (int i = 0; i < 8; i++) { r += acc[i]; r[0] += v[i] + (float)d[i]; }
    out[blockIdx.x * 256 + threadIdx.x] = r[0] + r[1] + r[2] + r[3];
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int MODE, int NV>
void run(const char *name)
{
    const int grid = 256, iters = 200;
    float *out; unsigned long long *cyc;
    (void)hipMalloc(&out, grid * 256 * 4); (void)hipMalloc(&cyc, grid * 4 * 8);
    for (int rep = 0; rep < 3; rep++) k<MODE, NV><<<grid, 256>>>(out, cyc, iters, 1.0f);
    (void)hipDeviceSynchronize();
    std::vector<unsigned long long> h(grid * 4);
    (void)hipMemcpy(h.data(), cyc, grid * 4 * 8, hipMemcpyDeviceToHost);
    double s = 0; for (auto v : h) s += (double)v;
    printf("%-58s %7.2f cycles per (MFMA + %d ops)\n", name, s / h.size() / iters / 8, NV);
    (void)hipFree(out); (void)hipFree(cyc);
}

int main()
{
    run<0, 0>("MFMA alone");
    run<9, 4>("no MFMA, 4 x v_fma_f32");
    run<1, 1>("MFMA + 1 v_fma_f32"); run<1, 2>("MFMA + 2 v_fma_f32"); run<1, 4>("MFMA + 4 v_fma_f32"); run<1, 8>("MFMA + 8 v_fma_f32");
    run<2, 1>("MFMA + 1 v_fma_f64"); run<2, 2>("MFMA + 2 v_fma_f64"); run<2, 4>("MFMA + 4 v_fma_f64");
    run<3, 1>("MFMA + 1 v_rcp_f64"); run<3, 2>("MFMA + 2 v_rcp_f64");
    run<4, 1>("MFMA + 1 (cvt_f64_f32 + add_f64)"); run<4, 2>("MFMA + 2 (cvt_f64_f32 + add_f64)");
    run<5, 1>("MFMA + 1 (cvt_f32_f64 + add_f32)"); run<5, 2>("MFMA + 2 (cvt_f32_f64 + add_f32)");
    run<6, 1>("MFMA + 1 (cvt_i32_f64 + cvt_f32_i32 + add_f32)"); run<6, 2>("MFMA + 2 (cvt_i32_f64 + cvt_f32_i32 + add_f32)");
    run<7, 1>("MFMA + 1 v_div_fixup_f64"); run<7, 2>("MFMA + 2 v_div_fixup_f64");
    run<8, 1>("MFMA + 1 v_rcp_f32"); run<8, 2>("MFMA + 2 v_rcp_f32"); run<8, 4>("MFMA + 4 v_rcp_f32");
    return 0;
}
