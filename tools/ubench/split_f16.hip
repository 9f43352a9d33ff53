// split_f16.hip -- checks behind the split-f16 arithmetic (an f32 product as 3 f16 MFMA products, f32 accumulate):
//   1. lane maps of v_mfma_f32_16x16x32_f16 and the D -> B identity over a PAIR of 16-row result tiles
//   2. f16 subnormal operands are not flushed by the MFMA
//   3. accuracy of a two-layer product against f64, next to v_mfma_f32_16x16x4_f32
//   4. rate of the 3-product group with the split VALU work in the gaps
// usage: ./split_f16
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ void split(float v, _Float16 &hi, _Float16 &lo)
{
    hi = (_Float16)v;
    lo = (_Float16)(v - (float)hi);
}

// One wave.  X [K][16 frames], W1 [32 hidden][K], W2 [16 outputs][32 hidden]:  H = W1 X (two 16-row tiles),
// O = W2 H with H taken from the accumulators (k-slot 8g + j <-> hidden (j < 4 ? 4g + j : 16 + 4g + j - 4)).
template <int K>
__global__ void two_layer(const float *X, const float *W1, const float *W2, float *H, float *O)
{
    const int lane = threadIdx.x, r = lane & 15, g = lane >> 4;
    f4 acc[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
    for (int s = 0; s < K / 32; s++) {
        h8 bh, bl, ah[2], al[2];
        for (int j = 0; j < 8; j++) {
            const int k = 32 * s + 8 * g + j;
            _Float16 hi, lo;
            split(X[k * 16 + r], hi, lo); bh[j] = hi; bl[j] = lo;
            for (int t = 0; t < 2; t++) { split(W1[(16 * t + r) * K + k], hi, lo); ah[t][j] = hi; al[t][j] = lo; }
        }
        for (int t = 0; t < 2; t++) {
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[t], bh, acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[t], bl, acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[t], bh, acc[t], 0, 0, 0);
        }
    }
    for (int t = 0; t < 2; t++)
        for (int i = 0; i < 4; i++) H[(16 * t + 4 * g + i) * 16 + r] = acc[t][i];     // D: row 4g + i, col r
    h8 sh, sl, wh, wl;
    for (int j = 0; j < 8; j++) {
        _Float16 hi, lo;
        split(acc[j >> 2][j & 3], hi, lo); sh[j] = hi; sl[j] = lo;
        const int hid = j < 4 ? 4 * g + j : 16 + 4 * g + (j - 4);
        split(W2[r * 32 + hid], hi, lo); wh[j] = hi; wl[j] = lo;
    }
    f4 o = {0, 0, 0, 0};
    o = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl, sh, o, 0, 0, 0);
    o = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, sl, o, 0, 0, 0);
    o = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, sh, o, 0, 0, 0);
    for (int i = 0; i < 4; i++) O[(4 * g + i) * 16 + r] = o[i];
}

template <int K>
__global__ void two_layer_f32(const float *X, const float *W1, const float *W2, float *H, float *O)
{
    const int lane = threadIdx.x, r = lane & 15, g = lane >> 4;
    f4 acc[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
    for (int s = 0; s < K / 4; s++)
        for (int t = 0; t < 2; t++)
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(W1[(16 * t + r) * K + 4 * s + g], X[(4 * s + g) * 16 + r], acc[t], 0, 0, 0);
    for (int t = 0; t < 2; t++)
        for (int i = 0; i < 4; i++) H[(16 * t + 4 * g + i) * 16 + r] = acc[t][i];
    f4 o = {0, 0, 0, 0};
    for (int t = 0; t < 2; t++)
        for (int i = 0; i < 4; i++) o = __builtin_amdgcn_mfma_f32_16x16x4f32(W2[r * 32 + 16 * t + 4 * g + i], acc[t][i], o, 0, 0, 0);
    for (int i = 0; i < 4; i++) O[(4 * g + i) * 16 + r] = o[i];
}

// subnormal operands: A = 2^-20 everywhere (f16 subnormal), B = 2^10: sum over 32 k = 32 * 2^-10
__global__ void subnormal(float *out)
{
    h8 a, b;
    for (int j = 0; j < 8; j++) { a[j] = (_Float16)9.5367431640625e-07f; b[j] = (_Float16)1024.0f; }
    f4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    if (threadIdx.x == 0) out[0] = c[0];
}

// rate: a hidden-loop-shaped stream: per step 12 MFMAs of layer 1 + sigmoid-like VALU work + split + 54 MFMAs of layer 2
template <bool VALU>
__global__ __launch_bounds__(256) void rate(float *out, int steps)
{
    const int lane = threadIdx.x & 63;
    h8 a, b;
    for (int j = 0; j < 8; j++) { a[j] = (_Float16)(0.001f * (lane + j)); b[j] = (_Float16)(0.002f * (lane - j)); }
    f4 acc[18], pre[4];
    for (int i = 0; i < 18; i++) acc[i] = (f4){0, 0, 0, 0};
    for (int i = 0; i < 4; i++) pre[i] = (f4){0.1f * lane, 0.2f, 0.3f, 0.4f};
    for (int s = 0; s < steps; s++) {
        h8 sh[2], sl[2];
        for (int f = 0; f < 2; f++)
            for (int j = 0; j < 8; j++) {
                float v = pre[2 * (j >> 2) + f][j & 3];
                if (VALU) {
                    const double t = 1512775.3951951856 * (double)v;
                    const unsigned hi = 1072632447u - (unsigned)__double2int_rz(-t);
                    v = __builtin_amdgcn_rcpf(1.0f + (float)__hiloint2double((int)hi, 0));
                }
                _Float16 h, l;
                split(v, h, l); sh[f][j] = h; sl[f][j] = l;
            }
        for (int i = 0; i < 4; i++) pre[i] = (f4){0.01f, 0.02f, 0.03f, 0.04f};
        for (int k = 0; k < 6; k++)
            for (int i = 0; i < 4; i++) {
                pre[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, pre[i], 0, 0, 0);
                pre[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b, a, pre[i], 0, 0, 0);
                pre[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, a, pre[i], 0, 0, 0);
            }
        for (int i = 0; i < 18; i++) {
            acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, sh[i & 1], acc[i], 0, 0, 0);
            acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, sl[i & 1], acc[i], 0, 0, 0);
            acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b, sh[i & 1], acc[i], 0, 0, 0);
        }
    }
    float r = 0;
    for (int i = 0; i < 18; i++) r += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int i = 0; i < 4; i++) r += pre[i][0];
    out[blockIdx.x * 256 + threadIdx.x] = r;
}

int main()
{
    constexpr int K = 192;
    std::vector<float> X(K * 16), W1(32 * K), W2(16 * 32), H(32 * 16), O(16 * 16), H32(32 * 16), O32(16 * 16);
    srand(7);
    auto rnd = [] { return (float)rand() / RAND_MAX * 2.0f - 1.0f; };
    for (auto &v : X) v = 3.0f * rnd();
    for (auto &v : W1) v = 0.2f * rnd();
    for (auto &v : W2) v = 0.5f * rnd();
    float *dX, *dW1, *dW2, *dH, *dO;
    CK(hipMalloc(&dX, X.size() * 4)); CK(hipMalloc(&dW1, W1.size() * 4)); CK(hipMalloc(&dW2, W2.size() * 4));
    CK(hipMalloc(&dH, H.size() * 4)); CK(hipMalloc(&dO, 65536 * 4));
    CK(hipMemcpy(dX, X.data(), X.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dW1, W1.data(), W1.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dW2, W2.data(), W2.size() * 4, hipMemcpyHostToDevice));
    two_layer<K><<<1, 64>>>(dX, dW1, dW2, dH, dO);
    CK(hipMemcpy(H.data(), dH, H.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(O.data(), dO, O.size() * 4, hipMemcpyDeviceToHost));
    two_layer_f32<K><<<1, 64>>>(dX, dW1, dW2, dH, dO);
    CK(hipMemcpy(H32.data(), dH, H.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(O32.data(), dO, O.size() * 4, hipMemcpyDeviceToHost));
    double eh = 0, eo = 0, eh32 = 0, eo32 = 0, mh = 0, mo = 0;
    std::vector<double> Hd(32 * 16);
    for (int h = 0; h < 32; h++)
        for (int f = 0; f < 16; f++) {
            double s = 0;
            for (int k = 0; k < K; k++) s += (double)W1[h * K + k] * X[k * 16 + f];
            Hd[h * 16 + f] = s;
            eh = fmax(eh, fabs(s - H[h * 16 + f])); eh32 = fmax(eh32, fabs(s - H32[h * 16 + f])); mh = fmax(mh, fabs(s));
        }
    for (int o = 0; o < 16; o++)
        for (int f = 0; f < 16; f++) {
            double s = 0;
            for (int h = 0; h < 32; h++) s += (double)W2[o * 32 + h] * Hd[h * 16 + f];
            eo = fmax(eo, fabs(s - O[o * 16 + f])); eo32 = fmax(eo32, fabs(s - O32[o * 16 + f])); mo = fmax(mo, fabs(s));
        }
    printf("two-layer product vs f64 (max |H| %.2f, max |O| %.2f):\n", mh, mo);
    printf("  split f16 (3 products): layer 1 max err %.3e   layer 2 %.3e\n", eh, eo);
    printf("  f32 MFMA              : layer 1 max err %.3e   layer 2 %.3e\n", eh32, eo32);
    subnormal<<<1, 64>>>(dO);
    float sub;
    CK(hipMemcpy(&sub, dO, 4, hipMemcpyDeviceToHost));
    printf("subnormal f16 operands: 32 * 2^-20 * 2^10 = %.6f (expected 0.031250; 0 = flushed)\n", sub);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int valu = 0; valu < 2; valu++) {
        const int steps = 2000;
        for (int rep = 0; rep < 2; rep++) {
            CK(hipEventRecord(e0));
            if (valu) rate<true><<<256, 256>>>(dO, steps); else rate<false><<<256, 256>>>(dO, steps);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
        }
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        const double mfmas = 126.0 * steps;
        printf("hidden-loop-shaped stream (%s): %.1f ns per step of 126 MFMAs = %.1f cycles per MFMA at 2.4 GHz; %.0f TFLOP/s f16\n",
               valu ? "sigmoid + split in the gaps" : "split only", ms * 1e6 / steps, ms * 1e-3 / mfmas * 2.4e9,
               mfmas * 16384 * 1024 / (ms * 1e-3) / 1e12);
    }
    return 0;
}
