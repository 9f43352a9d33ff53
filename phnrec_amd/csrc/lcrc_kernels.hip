// lcrc_kernels.hip -- the fused LCRC posterior kernel for gfx950 (MI355X).
//
// One launch computes, for every frame r of a batch of utterances,
//     post[r] = merger( ln band0(proj_L(ctx_r)) | ln band1(proj_R(ctx_r)) )
// i.e. everything Traps::CalcFeatures does per frame (traps.cpp:470-516):
//   AddVectorToBEMatrix           traps.cpp:180-219  -> clamped window gather from an LDS-staged mel tile
//   CalcInputFeaturesForBandNets  traps.cpp:285-343  -> window * DCT projection on the VALU
//   NeuralNet::Forward x3         nn.cpp:872-899     -> f32 MFMA (v_mfma_f32_16x16x4_f32) with fused
//                                                       normalise / bias / FEXP sigmoid / FEXP softmax
//   CalcInputFeaturesForMerger    traps.cpp:435-461  -> ln() + merger normalisation straight into the
//                                                       merger's operand image in LDS
//
// Geometry.  A workgroup owns kBM = 32 consecutive frames (two 16-frame MFMA
// column tiles) and runs the three nets one after the other; its NW waves split
// the HIDDEN dimension of each net.  Both products are computed transposed so
// that no data has to change lanes between them:
//   layer 1:  S^T[16 hidden x 16 frames] = W1[16 x K] . X^T          A = W1 fragment (HBM/L2, pre-packed)
//                                                                    B = X fragment  (LDS)
//   layer 2:  O^T[16 out x 16 frames]   += W2[16 x 16] . sig(S^T)     A = W2 fragment (HBM/L2, pre-packed)
//                                                                    B = the layer-1 accumulator itself
// The D layout of v_mfma_f32_16x16x4_f32 (row = 4*(lane>>4)+reg, col = lane&15) is
// exactly its B layout for k-slot (lane>>4) if register `reg` is used for k-step
// `reg`; so sig(S^T) feeds layer 2 from registers.  Hidden activations (M x 1500
// floats per net) therefore never exist in LDS or HBM.  Weights are streamed
// from L2/Infinity Cache as whole 1-KiB wave loads (host pre-packs them in
// fragment order), prefetched one hidden tile ahead.  At the end of a net the
// NW partial O^T tiles are folded through LDS, softmax runs on all threads,
// and the result becomes the next net's B image (band nets) or is stored to HBM
// as whole contiguous rows (merger).
//
// Arithmetic contract (tests/: <= 1e-4 max-abs per frame vs the reference):
//   * exp is the reference's FEXP bit trick, bit-emulated (fexp.h:14-21), incl.
//     x86's out-of-range cvttsd2si result; sigmoid is evaluated in f64 like the
//     reference's expression (fexp.h:33-38);
//   * products accumulate from the bias in ascending k (MFMA = f32 fma chain);
//     layer 2 is split over waves, then folded in a fixed order: deterministic;
//   * projection and normalisation use unfused f32 mul/add in the reference's
//     order (file is compiled with -ffp-contract=off).
#include <hip/hip_runtime.h>

#include <cfloat>
#include <climits>

#include "lcrc_dev.h"

namespace phnrec {

typedef float f4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f4 mfma16x16x4(float a, float b, f4 c)
{
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// ---- FEXP (fexp.h:14-21) ----------------------------------------------------------
// hi32 = (int)(2^20/ln2 * y) + (1072693248 - 60801); lo32 = 0; reinterpret as double.
// (int) is x86 cvttsd2si: INT_MIN when the product is >= 2^31 or NaN.  v_cvt_i32_f64
// saturates instead, so that one case is patched (y >= 2^11*ln2 = 1419.565...).
__device__ __forceinline__ double fexp_d(float y)
{
    const double a = 1048576.0 / 0.69314718055994530942;
    double t = a * (double)y;
    int i = __double2int_rz(t);
    if (!(t < 2147483648.0)) i = INT_MIN;
    unsigned hi = (unsigned)i + 1072632447u;
    return __hiloint2double((int)hi, 0);
}

__device__ __forceinline__ float fexp_f(float y) { return (float)fexp_d(y); }

// fexp_sigmoid (fexp.h:33-38): the macro yields a double, so 1.0f + .. and 1.0f / ..
// are double operations; one rounding to float.
__device__ __forceinline__ float fexp_sigmoid(float x)
{
    return (float)(1.0 / (1.0 + fexp_d(-x)));
}

// ---- one MLP on the workgroup's 32 frames ------------------------------------------
// XF: LDS image of the normalised input, [f][kq][lane] float4 where element j of
//     lane l holds X[frame 16f + (l&15)][k = 16kq + 4j + (l>>4)].
// On return dense[frame][o] (row stride 16*n_ot floats, aliases slab 1) holds the
// softmax output for all 32 frames; the caller must __syncthreads() before the
// LDS regions are reused.
template <int KS, int NOT, int NW, bool EXACT>
__device__ __forceinline__ void run_net(const NetDev &nd, const f4 *__restrict__ XF,
                                        f4 *__restrict__ slab, int n_ot_slab, int lane, int wave)
{
    constexpr int NKQ = (KS + 3) / 4;
    const int ks = EXACT ? KS : nd.ksteps;
    const int nkq = EXACT ? NKQ : nd.nkq;
    const int n_ot = EXACT ? NOT : nd.n_ot;
    const int g = lane >> 4;

    // layer-2 accumulators: acc[ot][f][rr] = O^T[16ot + 4g + rr][16f + (lane&15)]
    f4 acc[NOT][2];
#pragma unroll
    for (int ot = 0; ot < NOT; ot++) {
        f4 b = {0.f, 0.f, 0.f, 0.f};
        if (wave == 0 && (EXACT || ot < n_ot))
            b = *reinterpret_cast<const f4 *>(nd.b2 + 16 * ot + 4 * g);   // PrepareBiases nn.cpp:857
        acc[ot][0] = b;
        acc[ot][1] = b;
    }

    const int tpw = (nd.nht + NW - 1) / NW;
    const int ht0 = wave * tpw;
    const int ht1 = min(nd.nht, ht0 + tpw);
    const f4 *w1 = reinterpret_cast<const f4 *>(nd.w1p) + lane;
    const f4 *w2 = reinterpret_cast<const f4 *>(nd.w2p) + lane;

    f4 a[NKQ];
    if (ht0 < ht1) {
#pragma unroll
        for (int kq = 0; kq < NKQ; kq++)
            if (EXACT || kq < nkq) a[kq] = w1[(size_t)(ht0 * nkq + kq) * 64];
    }

    for (int ht = ht0; ht < ht1; ht++) {
        // layer-2 weights of this tile: in flight during layer 1
        f4 w[NOT];
#pragma unroll
        for (int ot = 0; ot < NOT; ot++)
            if (EXACT || ot < n_ot) w[ot] = w2[(size_t)(ht * n_ot + ot) * 64];

        // layer 1, bias first (nn.cpp:883-884): pre[f][r] = S^T[16ht + 4g + r][16f + (lane&15)]
        const f4 bias = *reinterpret_cast<const f4 *>(nd.b1 + 16 * ht + 4 * g);
        f4 p0 = bias, p1 = bias;
#pragma unroll
        for (int kq = 0; kq < NKQ; kq++) {
            if (EXACT || kq < nkq) {
                const f4 x0 = XF[kq * 64 + lane];
                const f4 x1 = XF[(nkq + kq) * 64 + lane];
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    if (EXACT ? (4 * kq + j < KS) : (4 * kq + j < ks)) {
                        p0 = mfma16x16x4(a[kq][j], x0[j], p0);
                        p1 = mfma16x16x4(a[kq][j], x1[j], p1);
                    }
                }
            }
        }
        // next tile's layer-1 weights: in flight during sigmoid + layer 2
        if (ht + 1 < ht1) {
#pragma unroll
            for (int kq = 0; kq < NKQ; kq++)
                if (EXACT || kq < nkq) a[kq] = w1[(size_t)((ht + 1) * nkq + kq) * 64];
        }
        // Sigmoid (nn.cpp:796-820).  Pad hidden units (>= n_hid) need no zeroing:
        // their layer-2 weights are packed as zeros.
        f4 s0, s1;
#pragma unroll
        for (int r = 0; r < 4; r++) {
            s0[r] = fexp_sigmoid(p0[r]);
            s1[r] = fexp_sigmoid(p1[r]);
        }
        // layer 2: k-slot g of step r is hidden unit 16ht + 4g + r on both operands
#pragma unroll
        for (int ot = 0; ot < NOT; ot++) {
            if (EXACT || ot < n_ot) {
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    acc[ot][0] = mfma16x16x4(w[ot][r], s0[r], acc[ot][0]);
                    acc[ot][1] = mfma16x16x4(w[ot][r], s1[r], acc[ot][1]);
                }
            }
        }
    }

    // ---- fold the NW partial tiles: top two waves down by two until one is left ----
    const int slab_f4 = 2 * n_ot_slab * 64;     // float4 per slab
#pragma unroll
    for (int top = NW; top > 1; top -= (top > 2 ? 2 : 1)) {
        const int nsrc = top > 2 ? 2 : 1;        // waves [top-nsrc, top) fold into [top-2*nsrc.. )
        const int src0 = top - nsrc, dst0 = src0 - nsrc;
        if (wave >= src0 && wave < top) {
            f4 *s = slab + (wave - src0) * slab_f4 + lane;
#pragma unroll
            for (int ot = 0; ot < NOT; ot++)
                if (EXACT || ot < n_ot) {
                    s[(ot * 2 + 0) * 64] = acc[ot][0];
                    s[(ot * 2 + 1) * 64] = acc[ot][1];
                }
        }
        __syncthreads();
        if (wave >= dst0 && wave < src0) {
            const f4 *s = slab + (wave - dst0) * slab_f4 + lane;
#pragma unroll
            for (int ot = 0; ot < NOT; ot++)
                if (EXACT || ot < n_ot) {
                    acc[ot][0] += s[(ot * 2 + 0) * 64];
                    acc[ot][1] += s[(ot * 2 + 1) * 64];
                }
        }
        __syncthreads();
    }

    // ---- logits -> dense[frame][o] (slab 1), softmax on all threads (nn.cpp:822-855) ----
    float *dense = reinterpret_cast<float *>(slab + slab_f4);
    const int os = 16 * n_ot_slab;
    if (wave == 0) {
#pragma unroll
        for (int ot = 0; ot < NOT; ot++)
            if (EXACT || ot < n_ot) {
#pragma unroll
                for (int f = 0; f < 2; f++)
                    *reinterpret_cast<f4 *>(dense + (16 * f + (lane & 15)) * os + 16 * ot + 4 * g) = acc[ot][f];
            }
    }
    __syncthreads();
    {
        constexpr int LPF = NW * 64 / kBM;       // lanes cooperating on one frame
        const int tid = wave * 64 + lane;
        const int frame = tid / LPF, part = tid % LPF;
        float *row = dense + frame * os;
        const int O = nd.n_out;
        float m = -FLT_MAX;
        for (int o = part; o < O; o += LPF) m = fmaxf(m, row[o]);
#pragma unroll
        for (int d = 1; d < LPF; d <<= 1) m = fmaxf(m, __shfl_xor(m, d));
        float sum = 0.0f;
        for (int o = part; o < O; o += LPF) {
            float e = fexp_f(row[o] - m);
            row[o] = e;
            sum += e;
        }
#pragma unroll
        for (int d = 1; d < LPF; d <<= 1) sum += __shfl_xor(sum, d);
        const float scale = 1.0f / sum;
        for (int o = part; o < O; o += LPF) row[o] *= scale;
    }
    __syncthreads();
}

// Scatter one value of a net-input row into the MFMA B image (see run_net).
__device__ __forceinline__ void xf_store(float *img, int nkq, int frame, int k, float v)
{
    const int f = frame >> 4, kq = k >> 4, j = (k >> 2) & 3, l = (frame & 15) + 16 * (k & 3);
    img[((f * nkq + kq) * 64 + l) * 4 + j] = v;
}

template <int KS1, int KSM, int NOT, int NW, bool EXACT>
__global__ __launch_bounds__(NW * 64) void lcrc_fused_kernel(const LcrcParams p)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NT = NW * 64;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nb = p.nbanks;
    const int nkq1 = EXACT ? (KS1 + 3) / 4 : p.net[0].nkq;
    const int nkqm = EXACT ? (KSM + 3) / 4 : p.net[2].nkq;
    const int n_ot = EXACT ? NOT : p.n_ot_slab;
    const LdsPlan lp = lcrc_lds_plan(nb, nkq1, nkqm, n_ot);

    float *melT = reinterpret_cast<float *>(smem + lp.mel);
    int *rowlo = reinterpret_cast<int *>(smem + lp.rowinfo);
    int *rowhi = rowlo + kBM;
    float *costab = reinterpret_cast<float *>(smem + lp.tabs);
    float *win = costab + 10 * 16;
    float *xf = reinterpret_cast<float *>(smem + lp.xf);
    float *gf = reinterpret_cast<float *>(smem + lp.gf);
    f4 *slab = reinterpret_cast<f4 *>(smem + lp.slab);

    const int r0 = blockIdx.x * kBM;
    const int tbase = r0 - kShift;

    // ---- stage 0: utterance bounds per frame, mel tile, tables, zeroed operand images ----
    if (tid < kBM) {
        const int r = min(r0 + tid, p.n_rows - 1);
        if (p.off == nullptr) {                 // one utterance of n_rows frames
            rowlo[tid] = 0;
            rowhi[tid] = p.n_rows - 1;
        } else {
            int lo = 0, hi = p.n_utts;          // largest u with off[u] <= r
            while (hi - lo > 1) {
                const int mid = (lo + hi) >> 1;
                if (p.off[mid] <= r) lo = mid; else hi = mid;
            }
            rowlo[tid] = p.off[lo];
            rowhi[tid] = p.off[lo + 1] - 1;
        }
    }
    for (int i = tid; i < kTileRows * nb; i += NT) {
        const int row = tbase + i / nb;
        melT[i] = (row >= 0 && row < p.n_rows) ? p.mel[(long)tbase * nb + i] : 0.0f;
    }
    for (int i = tid; i < 10 * 16 + 2 * 16; i += NT)
        costab[i] = i < 160 ? p.costab[i] : p.win[i - 160];
    {
        f4 *z = reinterpret_cast<f4 *>(xf);
        const int n = (int)((lp.slab - lp.xf) / 16);    // xf and gf are adjacent
        const f4 zero = {0.f, 0.f, 0.f, 0.f};
        for (int i = tid; i < n; i += NT) z[i] = zero;
    }
    __syncthreads();

    // ---- stage 1: window * DCT projection + input normalisation (traps.cpp:285-343,
    //      dspc.h:107-112,206-233, nn.cpp:702-716) ----
    {
        const NetDev &n0 = p.net[0];
        const int K = n0.n_inp;                  // nbanks * 11
        const int items = 2 * nb * kBM;
        for (int it = tid; it < items; it += NT) {
            const int i = it % kBM;
            const int b = (it / kBM) % nb;
            const int n = it / (kBM * nb);
            const int r = min(r0 + i, p.n_rows - 1);
            const int lo = rowlo[i], hi = rowhi[i];
            float xw[kHalf];
#pragma unroll
            for (int j = 0; j < kHalf; j++) {
                int s = r - kShift + n * kShift + j;
                s = max(lo, min(hi, s));
                xw[j] = melT[(s - tbase) * nb + b] * win[n * kHalf + j];
            }
            const NetDev &nd = p.net[n];
            float *img = xf + (size_t)n * (2 * nkq1 * 256);
            float *dbg = n == 0 ? p.dbg_in0 : p.dbg_in1;
            float sum = 0.0f;
#pragma unroll
            for (int j = 0; j < kHalf; j++) sum += xw[j];
            sum *= p.normc;                                          // CalcC0
            {
                const int k = b * kNCoef;
                if (dbg && r0 + i < p.n_rows) dbg[(size_t)(r0 + i) * K + k] = sum;
                float v = sum - nd.mean[k];
                v *= nd.dev[k];
                xf_store(img, nkq1, i, k, v);
            }
            for (int c = 0; c < kNCoef - 1; c++) {                   // sDCT
                float acc = 0.0f;
#pragma unroll
                for (int j = 0; j < kHalf; j++) acc += xw[j] * costab[c * 16 + j];
                acc *= p.normc;
                const int k = b * kNCoef + 1 + c;
                if (dbg && r0 + i < p.n_rows) dbg[(size_t)(r0 + i) * K + k] = acc;
                float v = acc - nd.mean[k];
                v *= nd.dev[k];
                xf_store(img, nkq1, i, k, v);
            }
        }
    }
    __syncthreads();

    // ---- stage 2: the two band nets; their ln() outputs build the merger's image ----
    const NetDev &nm = p.net[2];
    float *dense = reinterpret_cast<float *>(slab + 2 * n_ot * 64);
    const int os = 16 * n_ot;
#pragma unroll 1
    for (int n = 0; n < 2; n++) {
        const NetDev &nd = p.net[n];
        run_net<KS1, NOT, NW, EXACT>(nd, reinterpret_cast<const f4 *>(xf) + (size_t)n * (2 * nkq1 * 64),
                                     slab, n_ot, lane, wave);
        const int O = nd.n_out;
        const int kofs = n * p.net[0].n_out;
        float *dp = n == 0 ? p.dbg_p0 : p.dbg_p1;
        for (int it = tid; it < kBM * O; it += NT) {
            const int i = it / O, o = it % O;
            const float q = dense[i * os + o];
            const float gl = q > 0.0f ? logf(q) : 0.0f;             // sLn dspc.h:155-160
            if (r0 + i < p.n_rows) {
                if (dp) dp[(size_t)(r0 + i) * O + o] = q;
                if (p.dbg_g) p.dbg_g[(size_t)(r0 + i) * nm.n_inp + kofs + o] = gl;
            }
            float v = gl - nm.mean[kofs + o];
            v *= nm.dev[kofs + o];
            xf_store(gf, nkqm, i, kofs + o, v);
        }
        __syncthreads();
    }

    // ---- stage 3: merger; posteriors leave as whole contiguous rows ----
    run_net<KSM, NOT, NW, EXACT>(nm, reinterpret_cast<const f4 *>(gf), slab, n_ot, lane, wave);
    {
        const int O = nm.n_out;
        const int rows = min(kBM, p.n_rows - r0);
        float *dst = p.post + (size_t)r0 * O;
        for (int it = tid; it < rows * O; it += NT) {
            const int i = it / O, o = it % O;
            dst[it] = dense[i * os + o];
        }
    }
}

// ---- variants --------------------------------------------------------------------------
// (layer-1 k-steps of the band nets, of the merger, output tiles) of the shipped systems:
// CZ 165->1500->138 / 276  HU ..->186 / 372  RU 165->1400->159 / 318  EN 253->500->120 / 240
namespace {

constexpr int kNW = 4;
constexpr int kGenKS1 = 64, kGenKSM = 104, kGenNOT = 13;   // generic: <= 23 banks, <= 208 outputs

struct Variant {
    const char *name;
    int ks1, ksm, n_ot;    // 0,0,0 = generic
    const void *fn;
};

#define LCRC_KERNEL(KS1, KSM, NOT, EX) \
    reinterpret_cast<const void *>(&lcrc_fused_kernel<KS1, KSM, NOT, kNW, EX>)

const Variant kVariants[] = {
    {"cz_42_69_9", 42, 69, 9, LCRC_KERNEL(42, 69, 9, true)},
    {"hu_42_93_12", 42, 93, 12, LCRC_KERNEL(42, 93, 12, true)},
    {"ru_42_80_10", 42, 80, 10, LCRC_KERNEL(42, 80, 10, true)},
    {"en_64_60_8", 64, 60, 8, LCRC_KERNEL(64, 60, 8, true)},
    {"generic", 0, 0, 0, LCRC_KERNEL(kGenKS1, kGenKSM, kGenNOT, false)},
};

const Variant *pick(const NetDev *nets)
{
    for (const Variant &v : kVariants)
        if (v.ks1 == nets[0].ksteps && v.ks1 == nets[1].ksteps && v.ksm == nets[2].ksteps &&
            v.n_ot == nets[0].n_ot && v.n_ot == nets[1].n_ot && v.n_ot == nets[2].n_ot)
            return &v;
    if (nets[0].ksteps <= kGenKS1 && nets[1].ksteps == nets[0].ksteps && nets[2].ksteps <= kGenKSM &&
        nets[0].n_ot <= kGenNOT && nets[1].n_ot <= kGenNOT && nets[2].n_ot <= kGenNOT)
        return &kVariants[sizeof kVariants / sizeof kVariants[0] - 1];
    return nullptr;
}

}  // namespace

const char *lcrc_variant_for(const NetDev *nets, int nbanks, unsigned *lds_bytes)
{
    const Variant *v = pick(nets);
    if (!v) return nullptr;
    const LdsPlan lp = lcrc_lds_plan(nbanks, nets[0].nkq, nets[2].nkq, lcrc_n_ot_slab(nets));
    if (lds_bytes) *lds_bytes = lp.total;
    if (lp.total > 160u * 1024u) return nullptr;
    return v->name;
}

hipError_t lcrc_launch(const LcrcParams &p, hipStream_t stream, const char **variant_name)
{
    const Variant *v = pick(p.net);
    if (!v) return hipErrorInvalidValue;
    const LdsPlan lp = lcrc_lds_plan(p.nbanks, p.net[0].nkq, p.net[2].nkq, lcrc_n_ot_slab(p.net));
    if (lp.total > 160u * 1024u) return hipErrorInvalidValue;
    if (variant_name) *variant_name = v->name;
    if (p.n_rows <= 0) return hipSuccess;
    // > 64 KiB of dynamic LDS has to be granted per function (and per device)
    hipError_t e = hipFuncSetAttribute(v->fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    const dim3 grid((p.n_rows + kBM - 1) / kBM), block(kNW * 64);
    LcrcParams args = p;
    args.n_ot_slab = lcrc_n_ot_slab(p.net);
    void *kargs[] = {&args};
    return hipLaunchKernel(v->fn, grid, block, kargs, lp.total, stream);
}

}  // namespace phnrec
