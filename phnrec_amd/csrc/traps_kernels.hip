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

// Any posteriors/length, and LCRC at any geometry (a length other than 31, add_c0 = false, another number of
// coefficients per band): the same features with run-time sizes, a thread per (frame, band), every value re-read from the
// L1-resident mel rows instead of a register array.  No shipped model takes this path (all are LCRC, 31, add_c0).
//   modes 0 / 1: as traps_features_kernel
//   mode 2 (traps.cpp:285-343): LC = taps [0, half), RC = taps [half - 1, 2 half - 1) of each band, times the half
//     context's window, then [C0,] DCT over `half` points.  The reference walks be_mat ([band][L], flat) with a stride
//     of 2 half - 1 per band (traps.cpp:296-306): the same thing for odd L, and for even L a walk that drifts one slot
//     per band across the band rows -- restated as it is: flat index q = b (2 half - 1) + j -> band q / L, tap q % L.
__global__ __launch_bounds__(256) void traps_features_general_kernel(const TrapsFeatParams p)
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
    const int L = p.trap_len, nb = p.nbanks;
    auto at = [&](int band, int tap) { return p.mel[(size_t)max(lo, min(hi, r - p.back + tap)) * nb + band]; };
    if (p.mode == 0) {
        float *o = p.out + ((size_t)b * p.n_rows + r) * L;
        for (int tap = 0; tap < L; tap++) {
            float v = at(b, tap);
            if (p.use_hamming) v = v * p.hamming[tap];            // sMultVect, traps.cpp:236-243
            o[tap] = v;
        }
        return;
    }
    const int c0 = p.add_c0 ? 1 : 0, n_dct = p.shift - c0;
    if (p.mode == 1) {
        float *o = p.out + (size_t)r * ((size_t)p.trap_bands * p.shift) + (size_t)b * p.shift;
        auto x = [&](int j) { const float v = at(b, j); return p.use_hamming ? v * p.hamming[j] : v; };
        if (c0) {                                // CalcC0 dspc.h:223-233
            float sum = 0.0f;
            for (int j = 0; j < L; j++) sum += x(j);
            sum *= p.normc;
            *o++ = sum;
        }
        for (int k = 0; k < n_dct; k++) {        // sDCT dspc.h:206-221: sequential f32 sum, then the scale
            float acc = 0.0f;
            const float *ct = p.costab + (size_t)k * L;
            for (int j = 0; j < L; j++) acc += x(j) * ct[j];
            acc *= p.normc;
            o[k] = acc;
        }
        return;
    }
    const int H = p.half, K = p.trap_bands * p.shift;
    for (int n = 0; n < 2; n++) {
        float *o = p.out + ((size_t)n * p.n_rows + r) * K + (size_t)b * p.shift;
        const float *w = p.win + n * H;
        auto x = [&](int j) {
            const int q = b * (2 * H - 1) + n * (H - 1) + j;
            float v = at(q / L, q % L);
            v = v * w[j];                                        // sMultVect, traps.cpp:312-313
            return v;
        };
        if (c0) {
            float sum = 0.0f;
            for (int j = 0; j < H; j++) sum += x(j);
            sum *= p.normc;
            *o++ = sum;
        }
        for (int k = 0; k < n_dct; k++) {
            float acc = 0.0f;
            const float *ct = p.costab + (size_t)k * H;
            for (int j = 0; j < H; j++) acc += x(j) * ct[j];
            acc *= p.normc;
            o[k] = acc;
        }
    }
}

// LDS: [mean | dev] (2 * 16 * nkq floats), B image [ft][nkq][64] float4, four slabs [n_ot][ft][64] float4, and
// for the fused 1BT_DCT input: mel tile [(16 ft + 30)][nbanks], row bounds [2][16 ft], DCT basis [shift][31],
// Hamming window [32]
struct MlpLds {
    unsigned nrm, xf, slab, tile, rowinfo, costab, hamming, total;
};
__host__ __device__ inline MlpLds mlp_lds_plan(int nkq, int n_ot, int ft, int dct_banks, int dct_shift)
{
    MlpLds l;
    unsigned o = 0;
    l.nrm = o;     o += 2u * 16u * nkq * 4u;
    l.xf = o;      o += (unsigned)ft * nkq * 1024u;
    l.slab = o;    o += 4u * (unsigned)ft * n_ot * 1024u;
    l.tile = o;    o += dct_banks ? lcrc_round16((16u * ft + 2u * kShift) * dct_banks * 4u) : 0u;
    l.rowinfo = o; o += dct_banks ? 2u * 16u * ft * 4u : 0u;
    l.costab = o;  o += dct_banks ? lcrc_round16((unsigned)dct_shift * kTrapLen * 4u) : 0u;
    l.hamming = o; o += dct_banks ? 32u * 4u : 0u;
    l.total = o;
    return l;
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

// One NeuralNet::Forward over 16*FT rows per workgroup.  EXACT: the k-steps / output tiles are the template
// values (the shapes of the shipped LCRC band classifiers, which a 1BT_DCT model over the same banks shares);
// otherwise run-time sizes inside a size class.
template <int KS, int NOT, int NW, bool BATCHED, int FT, bool EXACT>
__global__ __launch_bounds__(NW * 64) void mlp_kernel(const MlpParams p)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NT = NW * 64, BM = 16 * FT;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const NetDev nd = BATCHED ? uniform_net(p.nets_dev + blockIdx.y) : p.net;
    const float *const in = BATCHED ? p.in + (size_t)blockIdx.y * p.in_net_stride : p.in;
    float *const out = BATCHED ? p.out + __builtin_amdgcn_readfirstlane(p.out_col[blockIdx.y]) : p.out;
    // the operand image is laid out with the CLASS's k-groups (run-time shapes: groups past the net's own hold zeros),
    // so that its addresses are compile-time offsets in the hidden loop (mlp_dev.h RingLoop); nkq_net: the net's own
    constexpr int nkq = (KS + 3) / 4;
    const int nkq_net = EXACT ? nkq : nd.nkq;
    const int n_ot = EXACT ? NOT : nd.n_ot, K = nd.n_inp, O = nd.n_out;
    const bool dct = !BATCHED && p.dct.mel != nullptr;
    const MlpLds L = mlp_lds_plan(nkq, n_ot, FT, dct ? p.dct.trap_bands : 0, dct ? p.dct.shift : 0);
    float *nrm = reinterpret_cast<float *>(smem + L.nrm);
    float *xf = reinterpret_cast<float *>(smem + L.xf);
    f4 *slab = reinterpret_cast<f4 *>(smem + L.slab);
    const int r0 = blockIdx.x * BM;

    for (int i = tid; i < 16 * nkq_net; i += NT) {
        nrm[i] = nd.mean[i];
        nrm[16 * nkq + i] = nd.dev[i];
    }
    {
        const f4 zero = {0.f, 0.f, 0.f, 0.f};
        f4 *z = reinterpret_cast<f4 *>(xf);
        for (int i = tid; i < FT * nkq * 64; i += NT) z[i] = zero;
    }
    if (dct) {
        // ---- fused 1BT_DCT input (AddVectorToBEMatrix + CalcInputFeaturesForBandNets, traps.cpp:180-283) ----
        const TrapsFeatParams &d = p.dct;
        const int nb = d.nbanks, tb = d.trap_bands, shift = d.shift;
        float *tile = reinterpret_cast<float *>(smem + L.tile);
        int *rowlo = reinterpret_cast<int *>(smem + L.rowinfo), *rowhi = rowlo + BM;
        float *ct = reinterpret_cast<float *>(smem + L.costab), *hm = reinterpret_cast<float *>(smem + L.hamming);
        const int tbase = r0 - kShift, trows = BM + 2 * kShift;
        for (int i = tid; i < trows * tb; i += NT) {        // bands >= trap_bands are never used (3BT has no DCT form)
            const int row = tbase + i / tb, b = i - (i / tb) * tb;
            tile[i] = (row >= 0 && row < d.n_rows) ? d.mel[(size_t)row * nb + b] : 0.0f;
        }
        const int n_dct = shift - (d.add_c0 ? 1 : 0);
        for (int i = tid; i < n_dct * kTrapLen; i += NT) ct[i] = d.costab[i];
        if (tid < kTrapLen) hm[tid] = d.hamming[tid];
        if (tid < BM) {
            const int r = min(r0 + tid, d.n_rows - 1);
            int lo = 0, hi = d.n_rows - 1;
            if (d.off) {                                     // largest u with off[u] <= r
                int a = 0, e = d.n_utts;
                while (e - a > 1) {
                    const int mid = (a + e) >> 1;
                    if (d.off[mid] <= r) a = mid; else e = mid;
                }
                lo = d.off[a];
                hi = d.off[a + 1] - 1;
            }
            rowlo[tid] = lo;
            rowhi[tid] = hi;
        }
        __syncthreads();
        if (shift <= 16) {
            // C0 / DCT of every band's trajectory as small MFMA products, like the LCRC projection (lcrc_kernels.hip stage 1):
            //   out[frame][c] = sum_tap (x[frame][tap] * hamming[tap]) * D[tap][c],   D[:, 0] = 1 (C0, with add_c0), then cos rows
            // A = 16 frames x 4 taps per step (gathered from the tile with clamped rows), B = the basis (8 registers per
            // lane, tap 31 = 0), taps summed in ascending order; an item = one (band, frame tile), dealt to the waves two at
            // a time (two independent chains of eight MFMAs).  The MFMA fuses each multiply-add, <= 1 ulp per term apart from
            // the features kernel's separate multiply and add: the fused and the three-launch form agree to ~1e-6, not
            // bit for bit.  (A workgroup's 480 (frame, band) pairs took ~4 us as 341 dependent LDS-fed FMAs per thread.)
            const int g = lane >> 4, c = lane & 15;
            const int c0 = d.add_c0 ? 1 : 0;
            float basis[8];
#pragma unroll
            for (int s8 = 0; s8 < 8; s8++) {
                const int tap = 4 * s8 + g;
                float v = 0.0f;
                if (tap < kTrapLen && c < shift) v = (c0 && c == 0) ? 1.0f : ct[(c - c0) * kTrapLen + tap];
                basis[s8] = v;
            }
            const float normc = d.normc;
            const int n_items = tb * FT;
            for (int it0 = 2 * wave; it0 < n_items; it0 += 2 * NW) {
                f4 acc[2];
                float x[2][8];
                int ib[2], ifr[2];
#pragma unroll
                for (int u = 0; u < 2; u++) {
                    const int it = min(it0 + u, n_items - 1);
                    ib[u] = it / FT; ifr[u] = it - ib[u] * FT;
                    const int i = 16 * ifr[u] + c;                 // this lane's frame as an A-operand row
                    const int r = min(r0 + i, d.n_rows - 1), lo = rowlo[i], hi = rowhi[i];
#pragma unroll
                    for (int s8 = 0; s8 < 8; s8++) {
                        const int tap = min(4 * s8 + g, kTrapLen - 1);
                        const int srow = max(lo, min(hi, r - kShift + tap));
                        float v = tile[(srow - tbase) * tb + ib[u]];
                        if (d.use_hamming) v = v * hm[tap];      // sMultVect, traps.cpp:236-243
                        x[u][s8] = v;
                    }
                    acc[u] = (f4){0.f, 0.f, 0.f, 0.f};
                }
#pragma unroll
                for (int s8 = 0; s8 < 8; s8++)
#pragma unroll
                    for (int u = 0; u < 2; u++) acc[u] = mfma16x16x4(x[u][s8], basis[s8], acc[u]);
#pragma unroll
                for (int u = 0; u < 2; u++) {
                    if (it0 + u < n_items && c < shift) {          // D layout: row = frame 4g + reg, col = c
                        const int k = ib[u] * shift + c;
                        const float mk = nrm[k], dk = nrm[16 * nkq + k];
#pragma unroll
                        for (int reg = 0; reg < 4; reg++) {
                            const float val = acc[u][reg] * normc;       // CalcC0 / sDCT scaling
                            float v = val - mk;                          // Normalize nn.cpp:702-716
                            v *= dk;
                            xf_store(xf, nkq, 16 * ifr[u] + 4 * g + reg, k, v);
                        }
                    }
                }
            }
        } else
        for (int pair = tid; pair < BM * tb; pair += NT) {
            const int i = pair / tb, b = pair - i * tb;
            const int r = min(r0 + i, d.n_rows - 1), lo = rowlo[i], hi = rowhi[i];
            float x[kTrapLen];
#pragma unroll
            for (int tap = 0; tap < kTrapLen; tap++) {
                const int srow = max(lo, min(hi, r - kShift + tap));
                float v = tile[(srow - tbase) * tb + b];
                if (d.use_hamming) v = v * hm[tap];          // sMultVect, traps.cpp:236-243
                x[tap] = v;
            }
            int k = b * shift;
            if (d.add_c0) {                                  // CalcC0 dspc.h:223-233
                float sum = 0.0f;
#pragma unroll
                for (int j = 0; j < kTrapLen; j++) sum += x[j];
                sum *= d.normc;
                float v = sum - nrm[k];                      // Normalize nn.cpp:702-716
                v *= nrm[16 * nkq + k];
                xf_store(xf, nkq, i, k, v);
                k++;
            }
            for (int c = 0; c < n_dct; c++, k++) {           // sDCT dspc.h:206-221: sequential f32 sum, then the scale
                float acc = 0.0f;
                const float *cc = ct + c * kTrapLen;
#pragma unroll
                for (int j = 0; j < kTrapLen; j++) acc += x[j] * cc[j];
                acc *= d.normc;
                float v = acc - nrm[k];
                v *= nrm[16 * nkq + k];
                xf_store(xf, nkq, i, k, v);
            }
        }
    } else {
        __syncthreads();
        // Row by row, a thread per input column (coalesced; no index division), eight rows' values requested before the
        // first is used: the 32 x 360 inputs of a 1BT merger cost 45 dependent load-divide-store rounds per thread before.
        constexpr int RB = 8;
        for (int k0 = 0; k0 < K; k0 += NT) {
            const int k = k0 + tid;
            const bool kin = k < K;
            const float mk = kin ? nrm[k] : 0.0f, dk = kin ? nrm[16 * nkq + k] : 0.0f;
            for (int i0 = 0; i0 < BM; i0 += RB) {
                float v[RB];
#pragma unroll
                for (int u = 0; u < RB; u++) {
                    const int r = r0 + i0 + u;
                    v[u] = (kin && r < p.n_rows) ? in[(size_t)r * p.in_ld + k] : 0.0f;
                }
#pragma unroll
                for (int u = 0; u < RB; u++) {
                    float w = v[u] - mk;                                 // Normalize nn.cpp:702-716
                    w *= dk;
                    if (kin) xf_store(xf, nkq, i0 + u, k, (r0 + i0 + u) < p.n_rows ? w : 0.0f);
                }
            }
        }
    }
    __syncthreads();

    float *outbuf = reinterpret_cast<float *>(slab);         // slab 0 is free again in the epilogue
    const bool transform = (p.out_func[0] | p.out_func[1] | p.out_be) != 0;
    auto epi = [&](int, int i, int o, float q, bool valid) {
        if (p.neg_log) {
            q = q > 0.0f ? logf(q) : 0.0f;                       // sLn dspc.h:155-160
            if (p.neg_log == 1) q = q * -1.0f;                   // 1BT / 3BT: sMultiplication(.., -1), traps.cpp:425
        } else if (transform) {
            q = soften(p.out_func[0], p.out_c[0], p.out_l[0], q);
            q = soften(p.out_func[1], p.out_c[1], p.out_l[1], q);
            if (p.out_be) q = __builtin_bit_cast(float, __builtin_bswap32(__builtin_bit_cast(unsigned, q)));
        }
        if (valid) outbuf[i * O + o] = q;
    };
    run_net<KS, NOT, NW, EXACT, FT, 1>(p, 0, &nd, reinterpret_cast<const f4 *>(xf), 0, slab, slab + 2 * FT * n_ot * 64, n_ot,
                                       lane, wave, per_value(epi));
    const int rows = min(BM, p.n_rows - r0);
    for (int o = tid; o < O; o += NT)                        // a thread per output column, row by row (coalesced)
        for (int i = 0; i < rows; i++) out[(size_t)(r0 + i) * p.out_ld + o] = outbuf[i * O + o];
}

// ---- 1BT / 3BT in ONE launch ------------------------------------------------------------------------------------------
// A workgroup owns 16*FT frames.  The band classifiers (31 -> H -> O_b, one per band) are tiny: as launches of their own
// (traps_features_kernel -> mlp_kernel with grid.y = band -> mlp_kernel for the merger) they cost 45 us + 10 us per 8192
// frames for 9 us of MFMA work, plus two round trips through HBM (15 MB of trajectories, 12 MB of merger inputs).
// Here each WAVE runs whole band classifiers by itself (bands b = wave, wave + 4, ...): it builds the band's normalised
// trajectory image from the LDS-staged mel tile, runs the hidden loop over all of the net's hidden tiles
// (hidden_range: the accumulators end up complete in its registers), takes them through a wave-private slab to the
// per-frame softmax, -ln() and the merger's normalisation, and scatters the values into the merger's operand image --
// no barrier until the merger starts, nothing leaves the CU.  Arithmetic: the band nets' products and FEXP as everywhere;
// their softmax sums four strided partials per frame, (P0 + P1) + (P2 + P3), whatever the tile size (16- and 32-frame
// workgroups agree bit for bit).  (traps.cpp:180-283,409-433)
struct Bt1Lds {
    unsigned nrm, gf, slab, tile, rowinfo, hamming, bimg, bnrm, total;
};
__host__ __device__ inline Bt1Lds bt1_lds_plan(int nkqm, int n_ot_m, int ft, int tile_banks)
{
    Bt1Lds l;
    unsigned o = 0;
    const unsigned n_ot = n_ot_m < 4 ? 4u : (unsigned)n_ot_m;     // the waves' band slabs (4 output tiles each) lie over the merger's
    l.nrm = o;     o += 2u * 16u * nkqm * 4u;
    l.gf = o;      o += (unsigned)ft * nkqm * 1024u;
    l.slab = o;    o += 4u * (unsigned)ft * n_ot * 1024u;
    l.tile = o;    o += lcrc_round16((16u * ft + 2u * kShift) * tile_banks * 4u);
    l.rowinfo = o; o += 2u * 16u * ft * 4u;
    l.hamming = o; o += 32u * 4u;
    l.bimg = o;    o += 4u * (unsigned)ft * 2u * 1024u;           // [wave][f][2 k-groups][64] float4
    l.bnrm = o;    o += (unsigned)tile_banks * 64u * 4u;          // [band][mean 32 | dev 32] of the band classifiers
    l.total = o;
    return l;
}

constexpr int kBandKS = 8, kBandNOT = 4;       // band classifiers: 31 inputs = 8 k-steps; two size classes: <= 32 / <= 64 outputs

template <int KSM, int NOTM, int BNOT, int NW, int FT>
__global__ __launch_bounds__(NW * 64) void traps_1bt_kernel(const MlpParams p)
{
    static_assert(BNOT <= kBandNOT, "band slabs are laid out for kBandNOT output tiles");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NT = NW * 64, BM = 16 * FT, nkqm = (KSM + 3) / 4;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const NetDev nm = p.net;
    const TrapsFeatParams &d = p.dct;
    const int tb = d.trap_bands, nb = d.nbanks, n_rows = d.n_rows;
    const int n_ot_m = nm.n_ot, O = nm.n_out;
    const Bt1Lds L = bt1_lds_plan(nkqm, n_ot_m, FT, tb);
    float *nrm = reinterpret_cast<float *>(smem + L.nrm);
    float *gf = reinterpret_cast<float *>(smem + L.gf);
    f4 *slab = reinterpret_cast<f4 *>(smem + L.slab);
    float *tile = reinterpret_cast<float *>(smem + L.tile);
    int *rowlo = reinterpret_cast<int *>(smem + L.rowinfo), *rowhi = rowlo + BM;
    float *hm = reinterpret_cast<float *>(smem + L.hamming);
    const int r0 = blockIdx.x * BM, tbase = r0 - kShift, trows = BM + 2 * kShift;

    // ---- prologue: merger normalisation vectors, zeroed merger image, mel tile, row bounds, Hamming window ----
    for (int i = tid; i < 16 * nm.nkq; i += NT) {
        nrm[i] = nm.mean[i];
        nrm[16 * nkqm + i] = nm.dev[i];
    }
    {
        const f4 zero = {0.f, 0.f, 0.f, 0.f};
        f4 *z = reinterpret_cast<f4 *>(gf);
        for (int i = tid; i < FT * nkqm * 64; i += NT) z[i] = zero;
    }
    for (int i = tid; i < trows * tb; i += NT) {                 // bands >= trap_bands are never used (3BT)
        const int row = tbase + i / tb, b = i - (i / tb) * tb;
        tile[i] = (row >= 0 && row < n_rows) ? d.mel[(size_t)row * nb + b] : 0.0f;
    }
    if (tid < 32) hm[tid] = (tid < kTrapLen && d.use_hamming) ? d.hamming[tid] : 1.0f;
    float *bnrm = reinterpret_cast<float *>(smem + L.bnrm);
    for (int i = tid; i < p.n_nets * 64; i += NT) {              // (mean / dev are padded to 32 floats per net)
        const NetDev *nb_ = p.nets_dev + (i >> 6);
        bnrm[i] = (i & 32) ? nb_->dev[i & 31] : nb_->mean[i & 31];
    }
    if (tid < BM) {
        const int r = min(r0 + tid, n_rows - 1);
        int lo = 0, hi = n_rows - 1;
        if (d.off) {                                             // largest u with off[u] <= r
            int a = 0, e = d.n_utts;
            while (e - a > 1) {
                const int mid = (a + e) >> 1;
                if (d.off[mid] <= r) a = mid; else e = mid;
            }
            lo = d.off[a];
            hi = d.off[a + 1] - 1;
        }
        rowlo[tid] = lo;
        rowhi[tid] = hi;
    }
    __syncthreads();

    // ---- the band classifiers, whole nets per wave ----
    {
        f4 *const bimg = reinterpret_cast<f4 *>(smem + L.bimg) + wave * (FT * 2 * 64);
        f4 *const bslab = slab + wave * (FT * kBandNOT * 64);
        float *const bs = reinterpret_cast<float *>(bslab);
        const float *mmean = nrm, *mdev = nrm + 16 * nkqm;
        const int c = lane & 15, g = lane >> 4;
        // this lane's source rows in the tile for its 2 k-groups x 4 taps per frame tile: tap k = 16 kq + 4 j + g
        int lo[FT], hi[FT], rc[FT];
#pragma unroll
        for (int f = 0; f < FT; f++) {
            const int i = 16 * f + c;
            lo[f] = rowlo[i]; hi[f] = rowhi[i];
            rc[f] = min(r0 + i, n_rows - 1) - kShift;
        }
        // softmax lanes: frame i = lane % BM, quarter h of the outputs: NH = 64 / BM lanes per frame, each carries 4 / NH of
        // the four strided partial sums
        constexpr int NH = 64 / BM, PPL = 4 / NH;
        const int si = lane % BM, sh = lane / BM;
        const int sbase = (((si >> 4) * 64) + (si & 15)) * 4;     // float index of (o = 0, frame si) in a slab tile row
        // (a) a band's normalised trajectories as its net's operand image (sMultVect traps.cpp:236-243, Normalize nn.cpp:702-716)
        auto build_image = [&](int b, f4 *img) {
            const float *bm_ = bnrm + b * 64;
#pragma unroll
            for (int f = 0; f < FT; f++)
#pragma unroll
                for (int kq = 0; kq < 2; kq++) {
                    f4 v;
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        const int k = 16 * kq + 4 * j + g;
                        const int kc = min(k, kTrapLen - 1);
                        const int srow = max(lo[f], min(hi[f], rc[f] + kc));
                        const float x = tile[(srow - tbase) * tb + b] * hm[kc];      // (window of ones without Hamming)
                        float w = x - bm_[kc];
                        w *= bm_[32 + kc];
                        v[j] = k < kTrapLen ? w : 0.0f;
                    }
                    img[(f * 2 + kq) * 64 + lane] = v;
                }
        };
        for (int b = wave; b < p.n_nets; b += NW) {
            const NetDev nd = uniform_net(p.nets_dev + b);
            const int mycol = __builtin_amdgcn_readfirstlane(p.out_col[b]);
            build_image(b, bimg);
            __builtin_amdgcn_wave_barrier();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            // (b) both layers on this wave alone: complete output tiles in registers
            f4 acc[BNOT][FT];
            hidden_range<kBandKS, BNOT, false, FT>(nd, bimg, 0, nd.nht, true, lane, acc);
            // (c) through the wave's slab to a per-frame view
            store_partial<BNOT, false, FT>(bslab, nd.n_ot, lane, acc);
            const int Ob = nd.n_out;
            __builtin_amdgcn_wave_barrier();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            // (d) softmax (nn.cpp:822-855), -ln (traps.cpp:424-425), the merger's normalisation, into its operand image.
            //     A lane's outputs o = sh + NH t sit in registers; element (o, frame si) of the slab: o = 16 ot + 4 g' + rr.
            //     Whole steps t past the net's outputs are skipped (a wave-uniform test).
            constexpr int MAXV = 16 * BNOT / NH;
            float v[MAXV];
            float m = -FLT_MAX;
#pragma unroll
            for (int t = 0; t < MAXV; t++) {
                v[t] = -FLT_MAX;
                if (NH * t < Ob) {
                    const int o = sh + NH * t;
                    const int oc = min(o, 16 * BNOT - 1);
                    const float x = bs[((oc >> 4) * FT * 64 + ((oc >> 2) & 3) * 16) * 4 + (oc & 3) + sbase];
                    v[t] = o < Ob ? x : -FLT_MAX;
                    m = fmaxf(m, v[t]);
                }
            }
            m = fmaxf(m, __shfl_xor(m, BM));
            if (NH == 4) m = fmaxf(m, __shfl_xor(m, 2 * BM));
            float ps[PPL];
#pragma unroll
            for (int q = 0; q < PPL; q++) ps[q] = 0.0f;
#pragma unroll
            for (int t = 0; t < MAXV; t++) {                      // partial (o mod 4) = sh + NH (t mod PPL), summed in order of o
                if (NH * t < Ob) {
                    const float e = fexp_nonpos_f(v[t] - m);
                    v[t] = (sh + NH * t) < Ob ? e : 0.0f;
                    ps[t % PPL] += v[t];
                }
            }
            float sum;
            if (NH == 4) {                               // lane h holds P_h: (P0 + P1) + (P2 + P3)
                const float t01 = ps[0] + __shfl_xor(ps[0], BM);
                sum = t01 + __shfl_xor(t01, 2 * BM);
            } else {                                     // lane h holds P_h and P_{h+2}
                const float t01 = ps[0] + __shfl_xor(ps[0], BM), t23 = ps[PPL - 1] + __shfl_xor(ps[PPL - 1], BM);
                sum = t01 + t23;
            }
            const float scale = 1.0f / sum;
#pragma unroll
            for (int t = 0; t < MAXV; t++) {
                if (NH * t < Ob) {
                    const int o = sh + NH * t;
                    float q = v[t] * scale;
                    q = (q > 0.0f ? logf(q) : 0.0f) * -1.0f;
                    const int k = min(mycol + o, 16 * nkqm - 1);
                    float w = q - mmean[k];
                    w *= mdev[k];
                    if (o < Ob) xf_store(gf, nkqm, si, k, w);
                }
            }
        }
    }
    __syncthreads();

    // ---- the merger on all four waves (as mlp_kernel) ----
    float *outbuf = reinterpret_cast<float *>(slab);
    const bool transform = (p.out_func[0] | p.out_func[1] | p.out_be) != 0;
    auto epi = [&](int, int i, int o, float q, bool valid) {
        if (transform) {
            q = soften(p.out_func[0], p.out_c[0], p.out_l[0], q);
            q = soften(p.out_func[1], p.out_c[1], p.out_l[1], q);
            if (p.out_be) q = __builtin_bit_cast(float, __builtin_bswap32(__builtin_bit_cast(unsigned, q)));
        }
        if (valid) outbuf[i * O + o] = q;
    };
    run_net<KSM, NOTM, NW, false, FT, 1>(p, 0, &nm, reinterpret_cast<const f4 *>(gf), 0, slab, slab + 2 * FT * n_ot_m * 64, n_ot_m,
                                         lane, wave, per_value(epi));
    const int rows = min(BM, n_rows - r0);
    for (int o = tid; o < O; o += NT)
        for (int i = 0; i < rows; i++) p.out[(size_t)(r0 + i) * p.out_ld + o] = outbuf[i * O + o];
}

bool mlp_supports(const NetDev &net)
{
    // (LDS: the largest size class's image, 16-frame workgroups, the fused 1BT_DCT input at its largest)
    return net.ksteps <= kMlpKS && net.n_ot <= kMlpNOT && mlp_lds_plan((kMlpKS + 3) / 4, net.n_ot, 1, 64, kTrapLen).total <= 160u * 1024u;
}

hipError_t traps_features_launch(const TrapsFeatParams &p, hipStream_t stream)
{
    if (p.n_rows <= 0) return hipSuccess;
    const long n = (long)p.n_rows * p.trap_bands;
    if (p.trap_len == kTrapLen && p.mode < 2)
        traps_features_kernel<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream>>>(p);
    else
        traps_features_general_kernel<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream>>>(p);
    return hipGetLastError();
}

namespace {

constexpr int kMlpNW = 4;
struct MlpVariant {
    const char *name;
    int ks, n_ot;            // exact k-steps / output tiles, or the class's maxima
    bool exact, batched;
    const void *fn[2];       // [FT - 1]; NULL = not built
};
#define MLP_FN(KS, NOT, B, FT, EX) reinterpret_cast<const void *>(&mlp_kernel<KS, NOT, kMlpNW, B, FT, EX>)
const MlpVariant kMlp[] = {
    // the band-classifier shapes of the shipped systems (32-frame workgroups; small launches use the classes)
    {"mlp_42_9", 42, 9, true, false, {nullptr, MLP_FN(42, 9, false, 2, true)}},
    {"mlp_42_12", 42, 12, true, false, {nullptr, MLP_FN(42, 12, false, 2, true)}},
    {"mlp_42_10", 42, 10, true, false, {nullptr, MLP_FN(42, 10, false, 2, true)}},
    {"mlp_64_8", 64, 8, true, false, {nullptr, MLP_FN(64, 8, false, 2, true)}},
    // size classes (k-steps, output tiles), first fit: every MFMA group of a class runs (zero fragments past the net's own
    // sizes), so a net costs its class's maxima
    {"mlp_le8_4", 8, 4, false, false, {MLP_FN(8, 4, false, 1, false), MLP_FN(8, 4, false, 2, false)}},
    {"mlp_le48_9", 48, 9, false, false, {MLP_FN(48, 9, false, 1, false), MLP_FN(48, 9, false, 2, false)}},
    {"mlp_le96_9", 96, 9, false, false, {MLP_FN(96, 9, false, 1, false), MLP_FN(96, 9, false, 2, false)}},
    {"mlp_le64", 64, kMlpNOT, false, false, {MLP_FN(64, kMlpNOT, false, 1, false), MLP_FN(64, kMlpNOT, false, 2, false)}},
    {"mlp_le128", 128, kMlpNOT, false, false, {MLP_FN(128, kMlpNOT, false, 1, false), MLP_FN(128, kMlpNOT, false, 2, false)}},
    {"mlp_le256", kMlpKS, kMlpNOT, false, false, {MLP_FN(kMlpKS, kMlpNOT, false, 1, false), MLP_FN(kMlpKS, kMlpNOT, false, 2, false)}},
    // many nets in one launch (grid.y = net): the band classifiers of 1BT / 3BT take 31 inputs
    // (32-frame workgroups were tried for them: 61 us against 45 us for the 15 band nets of a 1BT model at 8192 frames --
    //  these launches are bound by per-workgroup fixed costs, and halving the workgroups halves what overlaps them)
    {"mlp_nets_le8_4", 8, 4, false, true, {MLP_FN(8, 4, true, 1, false), nullptr}},
    {"mlp_nets_le64", 64, kMlpNOT, false, true, {MLP_FN(64, kMlpNOT, true, 1, false), nullptr}},
};
constexpr int kNMlp = sizeof kMlp / sizeof kMlp[0];

}  // namespace

hipError_t mlp_launch(const MlpParams &p, hipStream_t stream, const char **variant)
{
    // batched form: the caller vouches for every net (mlp_supports) and passes the maxima in net.{ksteps,nkq,n_ot}
    if (!mlp_supports(p.net)) return hipErrorInvalidValue;
    if (p.n_rows <= 0 || (p.nets_dev && p.n_nets <= 0)) return hipSuccess;
    const bool batched = p.nets_dev != nullptr;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    static std::atomic<int> cus[64] = {};
    const bool cached = dev >= 0 && dev < 64;
    int n_cu = cached ? cus[dev].load() : 0;
    if (n_cu == 0) {
        e = hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev);
        if (e != hipSuccess) return e;
        if (cached) cus[dev] = n_cu;
    }
    // 32-frame workgroups (every weight fragment serves two frame tiles) once they fill at least half of the CUs
    int ft = p.tile_frames == 16 ? 1 : p.tile_frames == 32 ? 2 : ((p.n_rows + 31) / 32 > n_cu / 2 ? 2 : 1);
    const int dct_banks = (!batched && p.dct.mel) ? p.dct.trap_bands : 0, dct_shift = dct_banks ? p.dct.shift : 0;
    auto pick = [&](int f) -> const MlpVariant * {
        for (const MlpVariant &c : kMlp) {
            if (c.batched != batched || !c.fn[f - 1]) continue;
            if (c.exact ? (c.ks == p.net.ksteps && c.n_ot == p.net.n_ot) : (p.net.ksteps <= c.ks && p.net.n_ot <= c.n_ot)) return &c;
        }
        return nullptr;
    };
    // LDS: the operand image has the VARIANT's k-groups (the class's for run-time shapes), see mlp_kernel
    auto lds_of = [&](const MlpVariant *c, int f) {
        return mlp_lds_plan((c->ks + 3) / 4, p.net.n_ot, f, dct_banks, dct_shift).total;
    };
    const MlpVariant *v = pick(ft);
    if (ft == 2 && (!v || lds_of(v, 2) > 160u * 1024u)) {
        ft = 1;
        v = pick(1);
    }
    if (!v || lds_of(v, ft) > 160u * 1024u) return hipErrorInvalidValue;   // (batched nets beyond 64 k-steps: band classifiers take 31 inputs)
    if (variant) *variant = v->name;
    const void *fn = v->fn[ft - 1];
    static std::atomic<bool> granted[kNMlp][2][64] = {};
    const int vi = (int)(v - kMlp);
    if (!cached || !granted[vi][ft - 1][dev]) {
        e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        if (cached) granted[vi][ft - 1][dev] = true;
    }
    MlpParams args = p;
    void *kargs[] = {&args};
    const int bm = 16 * ft;
    return hipLaunchKernel(fn, dim3((p.n_rows + bm - 1) / bm, batched ? p.n_nets : 1), dim3(kMlpNW * 64), kargs,
                           lds_of(v, ft), stream);
}

// ---- 1BT / 3BT fused launch: merger size classes as mlp_kernel's (k-steps, output tiles), first fit ----
namespace {
struct Bt1Variant {
    const char *name;
    int ks, n_ot, band_n_ot;
    const void *fn[2];       // [FT - 1]
};
#define BT1_FN(KS, NOT, BNOT, FT) reinterpret_cast<const void *>(&traps_1bt_kernel<KS, NOT, BNOT, kMlpNW, FT>)
const Bt1Variant kBt1[] = {     // merger class (k-steps, output tiles) x band class (output tiles), first fit
    {"traps_1bt_le96_9_b2", 96, 9, 2, {BT1_FN(96, 9, 2, 1), BT1_FN(96, 9, 2, 2)}},
    {"traps_1bt_le96_9_b4", 96, 9, 4, {BT1_FN(96, 9, 4, 1), BT1_FN(96, 9, 4, 2)}},
    {"traps_1bt_le128_13_b2", 128, kMlpNOT, 2, {BT1_FN(128, kMlpNOT, 2, 1), BT1_FN(128, kMlpNOT, 2, 2)}},
    {"traps_1bt_le128_13_b4", 128, kMlpNOT, 4, {BT1_FN(128, kMlpNOT, 4, 1), BT1_FN(128, kMlpNOT, 4, 2)}},
    {"traps_1bt_le256_13_b4", kMlpKS, kMlpNOT, 4, {BT1_FN(kMlpKS, kMlpNOT, 4, 1), nullptr}},
};
constexpr int kNBt1 = sizeof kBt1 / sizeof kBt1[0];
}  // namespace

// p.net = the merger, p.nets_dev / p.out_col / p.n_nets = the band classifiers (p.lds_n_ot = their largest n_ot),
// p.dct = mel / offsets / banks / Hamming window (mode 0).  hipErrorNotSupported: no class holds this model (the caller
// falls back to the three-launch form).
hipError_t traps_1bt_launch(const MlpParams &p, hipStream_t stream, const char **variant)
{
    if (p.n_rows <= 0) return hipSuccess;
    if (!p.nets_dev || p.n_nets <= 0 || p.n_nets > 64 || p.lds_n_ot > kBandNOT || p.lds_nkq > 2) return hipErrorNotSupported;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    int n_cu = 0;
    e = hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev);
    if (e != hipSuccess) return e;
    int ft = p.tile_frames == 16 ? 1 : p.tile_frames == 32 ? 2 : ((p.n_rows + 31) / 32 > n_cu / 2 ? 2 : 1);
    auto pick = [&](int f) -> const Bt1Variant * {
        for (const Bt1Variant &c : kBt1)
            if (c.fn[f - 1] && p.net.ksteps <= c.ks && p.net.n_ot <= c.n_ot && p.lds_n_ot <= c.band_n_ot &&
                bt1_lds_plan((c.ks + 3) / 4, p.net.n_ot, f, p.dct.trap_bands).total <= 160u * 1024u) return &c;
        return nullptr;
    };
    const Bt1Variant *v = pick(ft);
    if (!v && ft == 2) { ft = 1; v = pick(1); }
    if (!v) return hipErrorNotSupported;
    if (variant) *variant = v->name;
    const void *fn = v->fn[ft - 1];
    static std::atomic<bool> granted[kNBt1][2][64] = {};
    const int vi = (int)(v - kBt1);
    const bool cached = dev >= 0 && dev < 64;
    if (!cached || !granted[vi][ft - 1][dev]) {
        e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        if (cached) granted[vi][ft - 1][dev] = true;
    }
    MlpParams args = p;
    void *kargs[] = {&args};
    const int bm = 16 * ft;
    return hipLaunchKernel(fn, dim3((p.n_rows + bm - 1) / bm), dim3(kMlpNW * 64), kargs,
                           bt1_lds_plan((v->ks + 3) / 4, p.net.n_ot, ft, p.dct.trap_bands).total, stream);
}

}  // namespace phnrec
