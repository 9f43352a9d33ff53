// Micro-benchmark (dev tool): TWO waves per SIMD.  What does a second wave on the same SIMD hide?
// Workgroup of 512 threads: waves 0-3 (role A) and 4-7 (role B) land pairwise on the four SIMDs.
// Each role runs one of: idle, MFMA only, VALU only, global loads only, LDS reads only, or the
// hidden loop's MIX (84 MFMAs + 11 1-KiB loads + 22 LDS reads + 64 VALU ops per iteration).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0)

enum { IDLE = 0, MFMA_ONLY, VALU_ONLY, LOAD_ONLY, LDS_ONLY, MIX };

template <int ROLE>
__device__ __forceinline__ unsigned long long body(const f4 *wts, const f4 *lds, int iters, int lane, float x, float *sink)
{
    f4 acc[4];
    for (int i = 0; i < 4; i++) acc[i] = (f4){x, x, x, x};
    float v[8];
    for (int i = 0; i < 8; i++) v[i] = x * (i + 1) + lane * 1e-3f;
    f4 w[11];
    for (int i = 0; i < 11; i++) w[i] = (f4){x, x + 1, x + 2, x + 3};
    f4 l[2] = {{x, x, x, x}, {x, x, x, x}};
    __builtin_amdgcn_sched_barrier(0);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_sched_barrier(0);
    for (int it = 0; it < iters; it++) {
        const f4 *t = wts + (size_t)(it & 63) * 11 * 64 + lane;
        if (ROLE == MFMA_ONLY || ROLE == MIX) {
#pragma unroll
            for (int q = 0; q < 11; q++) {
                if (ROLE == MIX) {
                    f4 nw = t[q * 64];                       // next iteration's fragment
                    l[0] = lds[q * 64 + lane];
                    l[1] = lds[(11 + q) * 64 + lane];
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        acc[0] = MFMA(w[q][j], l[0][j], acc[0]);
                        acc[1] = MFMA(w[q][j], l[1][j], acc[1]);
                    }
                    w[q] = nw;
                } else {
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        acc[0] = MFMA(w[q][j], l[0][j], acc[0]);
                        acc[1] = MFMA(w[q][j], l[1][j], acc[1]);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (ROLE == MIX) {
#pragma unroll
                for (int s = 0; s < 8; s++)
#pragma unroll
                    for (int i = 0; i < 8; i++) v[i] = __builtin_fmaf(v[i], 1.0001f, 0.5f);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (ROLE == VALU_ONLY) {
#pragma unroll
            for (int s = 0; s < 8; s++)
#pragma unroll
                for (int i = 0; i < 8; i++) v[i] = __builtin_fmaf(v[i], 1.0001f, 0.5f);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (ROLE == LOAD_ONLY) {
#pragma unroll
            for (int q = 0; q < 11; q++) w[q] += t[q * 64];
            __builtin_amdgcn_sched_barrier(0);
        }
        if (ROLE == LDS_ONLY) {
#pragma unroll
            for (int q = 0; q < 11; q++) { acc[2] += lds[q * 64 + lane]; acc[3] += lds[(11 + q) * 64 + lane]; }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    f4 r = acc[0] + acc[1] + acc[2] + acc[3];
    for (int i = 0; i < 11; i++) r += w[i];
    for (int i = 0; i < 8; i++) r[0] += v[i];
    *sink = r[0] + r[1] + r[2] + r[3];
    return t1 - t0;
}

template <int RA, int RB>
__global__ __launch_bounds__(512) void k(const f4 *wts, float *out, unsigned long long *cyc, int iters, float x)
{
    __shared__ f4 lds[22 * 64];
    for (int i = threadIdx.x; i < 22 * 64; i += 512) lds[i] = (f4){x, x, x, x};
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float sink = 0.f;
    unsigned long long c = 0;
    if (wave < 4) { if (RA != IDLE) c = body<RA>(wts, lds, iters, lane, x, &sink); }
    else          { if (RB != IDLE) c = body<RB>(wts, lds, iters, lane, x, &sink); }
    out[blockIdx.x * 512 + threadIdx.x] = sink;
    if (lane == 0) cyc[blockIdx.x * 8 + wave] = c;
}

template <int RA, int RB>
void run(const char *name, const f4 *wts)
{
    const int grid = 256, iters = 200;
    float *out; unsigned long long *cyc;
    (void)hipMalloc(&out, grid * 512 * 4); (void)hipMalloc(&cyc, grid * 8 * 8);
    for (int rep = 0; rep < 3; rep++) k<RA, RB><<<grid, 512>>>(wts, out, cyc, iters, 1.0f);
    (void)hipDeviceSynchronize();
    std::vector<unsigned long long> h(grid * 8);
    (void)hipMemcpy(h.data(), cyc, grid * 8 * 8, hipMemcpyDeviceToHost);
    double a = 0, b = 0;
    for (int g = 0; g < grid; g++) for (int w = 0; w < 8; w++) (w < 4 ? a : b) += (double)h[g * 8 + w];
    printf("%-44s A %8.1f  B %8.1f cycles per iteration\n", name, a / (grid * 4) / iters, b / (grid * 4) / iters);
    (void)hipFree(out); (void)hipFree(cyc);
}

int main()
{
    f4 *wts;
    const size_t n = 64 * 11 * 64;
    (void)hipMalloc(&wts, n * sizeof(f4));
    (void)hipMemset(wts, 0, n * sizeof(f4));
    printf("iteration = 88 MFMA | 64 v_fma_f32 | 11 1-KiB global loads | 22 1-KiB LDS reads | MIX = all of them\n");
    run<MFMA_ONLY, IDLE>("A = MFMA, B idle", wts);
    run<MFMA_ONLY, MFMA_ONLY>("A = MFMA, B = MFMA", wts);
    run<VALU_ONLY, IDLE>("A = VALU, B idle", wts);
    run<MFMA_ONLY, VALU_ONLY>("A = MFMA, B = VALU", wts);
    run<LOAD_ONLY, IDLE>("A = loads, B idle", wts);
    run<MFMA_ONLY, LOAD_ONLY>("A = MFMA, B = loads", wts);
    run<LDS_ONLY, IDLE>("A = LDS, B idle", wts);
    run<MFMA_ONLY, LDS_ONLY>("A = MFMA, B = LDS", wts);
    run<MIX, IDLE>("A = MIX, B idle", wts);
    run<MIX, MIX>("A = MIX, B = MIX", wts);
    return 0;
}
