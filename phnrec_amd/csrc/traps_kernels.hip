// traps_kernels.hip -- the 1BT_DCT / 1BT / 3BT variants of Traps (SURVEY.md 8 "next" row f4).
//
// No shipped model uses them (1BT_DCT is only the schema default, srec.cpp:69), so they are composed from
// two general kernels instead of one fused kernel per system:
//   traps_features_kernel   AddVectorToBEMatrix + CalcInputFeaturesForBandNets for these systems
//                           (traps.cpp:180-283): clamped 31-frame trajectories per band, optional Hamming
//                           window, and for 1BT_DCT the C0 / DCT projection in the reference's order
//   mlp_kernel              NeuralNet::Forward (nn.cpp:872-899) of ONE net over rows of a matrix in HBM, on
//                           the same MFMA machinery as the fused LCRC kernel (mlp_dev.h: run_net with
//                           run-time sizes); the epilogue either stores -ln(p) straight into the merger's
//                           input matrix (CalcInputFeaturesForMerger, traps.cpp:409-433) or applies the
//                           posterior writer path's softening / byte order.
#include <hip/hip_runtime.h>

#include <atomic>

#include "lcrc_dev.h"
#include "mlp_dev.h"

namespace phnrec {

__global__ __launch_bounds__(256) void traps_features_kernel(const TrapsFeatParams p)
{
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    const int r = (int)(idx / p.trap_bands), b = (int)(idx % p.trap_bands);
    if (r >= p.n_rows) return;
    int lo = 0, hi = p.n_rows - 1;
    if (p.off) {                                 // largest u with off[u] <= r
        int a = 0, e = p.n_utts;
        while (e - a > 1) {
            const int mid = (a + e) >> 1;
            if (p.off[mid] <= r) a = mid; else e = mid;
        }
        lo = p.off[a];
        hi = p.off[a + 1] - 1;
    }
    float x[kTrapLen];
#pragma unroll
    for (int tap = 0; tap < kTrapLen; tap++) {
        const int s = max(lo, min(hi, r - kShift + tap));
        float v = p.mel[(size_t)s * p.nbanks + b];
        if (p.use_hamming) v = v * p.hamming[tap];            // sMultVect, traps.cpp:236-243
        x[tap] = v;
    }
    if (p.mode == 0) {
        float *o = p.out + ((size_t)b * p.n_rows + r) * kTrapLen;
#pragma unroll
        for (int tap = 0; tap < kTrapLen; tap++) o[tap] = x[tap];
        return;
    }
    float *o = p.out + (size_t)r * ((size_t)p.trap_bands * p.shift) + (size_t)b * p.shift;
    int n_out = p.shift;
    if (p.add_c0) {                              // CalcC0 dspc.h:223-233
        float sum = 0.0f;
#pragma unroll
        for (int j = 0; j < kTrapLen; j++) sum += x[j];
        sum *= p.normc;
        *o++ = sum;
        n_out = p.shift - 1;
    }
    for (int k = 0; k < n_out; k++) {            // sDCT dspc.h:206-221: sequential f32 sum, then the scale
        float acc = 0.0f;
        const float *ct = p.costab + k * kTrapLen;
#pragma unroll
        for (int j = 0; j < kTrapLen; j++) acc += x[j] * ct[j];
        acc *= p.normc;
        o[k] = acc;
    }
}

// LDS: [mean | dev] (2 * 16 * nkq floats), B image [nkq][64] float4, four slabs [n_ot][64] float4
__host__ __device__ inline unsigned mlp_lds_bytes(int nkq, int n_ot)
{
    return 2u * 16u * nkq * 4u + (unsigned)nkq * 1024u + 4u * (unsigned)n_ot * 1024u;
}

// A NetDev read from device memory at a wave-uniform address, moved into SGPRs field by field: the weight
// pointers must be scalar for the hidden loop's scalar-base addressing.
template <typename T>
__device__ __forceinline__ T uniform_ptr(T v)
{
    const unsigned long long u = reinterpret_cast<unsigned long long>(v);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u), hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
    return reinterpret_cast<T>(((unsigned long long)hi << 32) | lo);
}

__device__ __forceinline__ NetDev uniform_net(const NetDev *g)
{
    NetDev n = *g;
    n.w1p = uniform_ptr(n.w1p); n.w2p = uniform_ptr(n.w2p);
    n.b1 = uniform_ptr(n.b1); n.b2 = uniform_ptr(n.b2); n.mean = uniform_ptr(n.mean); n.dev = uniform_ptr(n.dev);
    n.n_inp = __builtin_amdgcn_readfirstlane(n.n_inp); n.n_hid = __builtin_amdgcn_readfirstlane(n.n_hid);
    n.n_out = __builtin_amdgcn_readfirstlane(n.n_out); n.ksteps = __builtin_amdgcn_readfirstlane(n.ksteps);
    n.nkq = __builtin_amdgcn_readfirstlane(n.nkq); n.nht = __builtin_amdgcn_readfirstlane(n.nht);
    n.n_ot = __builtin_amdgcn_readfirstlane(n.n_ot);
    return n;
}

template <int KS, int NOT, int NW, bool BATCHED>
__global__ __launch_bounds__(NW * 64) void mlp_kernel(const MlpParams p)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NT = NW * 64, BM = 16;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const NetDev nd = BATCHED ? uniform_net(p.nets_dev + blockIdx.y) : p.net;
    const float *const in = BATCHED ? p.in + (size_t)blockIdx.y * p.in_net_stride : p.in;
    float *const out = BATCHED ? p.out + __builtin_amdgcn_readfirstlane(p.out_col[blockIdx.y]) : p.out;
    const int nkq = nd.nkq, n_ot = nd.n_ot, K = nd.n_inp, O = nd.n_out;
    float *nrm = reinterpret_cast<float *>(smem);
    float *xf = nrm + 2 * 16 * nkq;
    f4 *slab = reinterpret_cast<f4 *>(xf + nkq * 256);
    const int r0 = blockIdx.x * BM;

    for (int i = tid; i < 16 * nkq; i += NT) {
        nrm[i] = nd.mean[i];
        nrm[16 * nkq + i] = nd.dev[i];
    }
    {
        const f4 zero = {0.f, 0.f, 0.f, 0.f};
        f4 *z = reinterpret_cast<f4 *>(xf);
        for (int i = tid; i < nkq * 64; i += NT) z[i] = zero;
    }
    __syncthreads();
    for (int idx = tid; idx < BM * K; idx += NT) {
        const int i = idx / K, k = idx - i * K;
        const int r = r0 + i;
        float v = r < p.n_rows ? in[(size_t)r * p.in_ld + k] : 0.0f;
        v = v - nrm[k];                                      // Normalize nn.cpp:702-716
        v *= nrm[16 * nkq + k];
        xf_store(xf, nkq, i, k, v);
    }
    __syncthreads();

    float *outbuf = reinterpret_cast<float *>(slab);         // slab 0 is free again in the epilogue
    const bool transform = (p.out_func[0] | p.out_func[1] | p.out_be) != 0;
    auto epi = [&](int, int i, int o, float q, bool valid) {
        if (p.neg_log) {
            q = (q > 0.0f ? logf(q) : 0.0f) * -1.0f;
        } else if (transform) {
            q = soften(p.out_func[0], p.out_c[0], p.out_l[0], q);
            q = soften(p.out_func[1], p.out_c[1], p.out_l[1], q);
            if (p.out_be) q = __builtin_bit_cast(float, __builtin_bswap32(__builtin_bit_cast(unsigned, q)));
        }
        if (valid) outbuf[i * O + o] = q;
    };
    run_net<KS, NOT, NW, false, 1, 1>(p, 0, &nd, reinterpret_cast<const f4 *>(xf), 0, slab, slab + 2 * n_ot * 64, n_ot,
                                      lane, wave, per_value(epi));
    const int rows = min(BM, p.n_rows - r0);
    for (int idx = tid; idx < rows * O; idx += NT) {
        const int i = idx / O, o = idx - i * O;
        out[(size_t)(r0 + i) * p.out_ld + o] = outbuf[idx];
    }
}

bool mlp_supports(const NetDev &net)
{
    return net.ksteps <= kMlpKS && net.n_ot <= kMlpNOT && mlp_lds_bytes(net.nkq, net.n_ot) <= 160u * 1024u;
}

hipError_t traps_features_launch(const TrapsFeatParams &p, hipStream_t stream)
{
    if (p.n_rows <= 0) return hipSuccess;
    const long n = (long)p.n_rows * p.trap_bands;
    traps_features_kernel<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream>>>(p);
    return hipGetLastError();
}

hipError_t mlp_launch(const MlpParams &p, hipStream_t stream)
{
    // batched form: the caller vouches for every net (mlp_supports) and passes the maxima in net.{ksteps,nkq,n_ot}
    if (!mlp_supports(p.net)) return hipErrorInvalidValue;
    if (p.n_rows <= 0 || (p.nets_dev && p.n_nets <= 0)) return hipSuccess;
    constexpr int NW = 4;
    // four size classes of the same kernel (k-steps, output tiles): the loops are unrolled over the class's
    // maxima, so a small net in a large class would step over mostly empty entries
    const int cls = (p.net.ksteps <= 8 && p.net.n_ot <= 4) ? 0 : p.net.ksteps <= 64 ? 1 : p.net.ksteps <= 128 ? 2 : 3;
    const bool batched = p.nets_dev != nullptr;
    const void *fn = batched ? (cls == 0 ? reinterpret_cast<const void *>(&mlp_kernel<8, 4, NW, true>)
                                         : reinterpret_cast<const void *>(&mlp_kernel<64, kMlpNOT, NW, true>))
                   : cls == 0 ? reinterpret_cast<const void *>(&mlp_kernel<8, 4, NW, false>)
                   : cls == 1 ? reinterpret_cast<const void *>(&mlp_kernel<64, kMlpNOT, NW, false>)
                   : cls == 2 ? reinterpret_cast<const void *>(&mlp_kernel<128, kMlpNOT, NW, false>)
                              : reinterpret_cast<const void *>(&mlp_kernel<kMlpKS, kMlpNOT, NW, false>);
    if (batched && cls >= 2) return hipErrorInvalidValue;    // band classifiers take 31 inputs
    static std::atomic<bool> granted[2][4][64] = {};
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 0 || dev >= 64 || !granted[batched][cls][dev]) {
        e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        if (dev >= 0 && dev < 64) granted[batched][cls][dev] = true;
    }
    MlpParams args = p;
    void *kargs[] = {&args};
    return hipLaunchKernel(fn, dim3((p.n_rows + 15) / 16, p.nets_dev ? p.n_nets : 1), dim3(NW * 64), kargs,
                           mlp_lds_bytes(p.net.nkq, p.net.n_ot), stream);
}

}  // namespace phnrec
