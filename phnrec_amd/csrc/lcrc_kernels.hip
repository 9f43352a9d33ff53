// lcrc_kernels.hip -- the fused LCRC posterior kernel for gfx950 (MI355X).
//
// One launch computes, for every frame r of a batch of utterances,
//     post[r] = merger( ln band0(proj_L(ctx_r)) | ln band1(proj_R(ctx_r)) )
// i.e. everything Traps::CalcFeatures does per frame (traps.cpp:470-516):
//   AddVectorToBEMatrix           traps.cpp:180-219  -> clamped window gather from an LDS-staged mel tile
//   CalcInputFeaturesForBandNets  traps.cpp:285-343  -> window * DCT projection as small MFMA products (stage 1)
//   NeuralNet::Forward x3         nn.cpp:872-899     -> f32 MFMA (v_mfma_f32_16x16x4_f32) with fused
//                                                       normalise / bias / FEXP sigmoid / FEXP softmax
//   CalcInputFeaturesForMerger    traps.cpp:435-461  -> ln() + merger normalisation straight into the
//                                                       merger's operand image in LDS
//
// Geometry.  A workgroup owns 16*FT consecutive frames (FT = 1 or 2 16-frame MFMA column
// tiles: lcrc_launch runs whole rounds as PAIRS of 16-frame workgroups per CU where two
// of them fit side by side -- every shipped shape --, one 32-frame workgroup per CU
// otherwise, and smaller launches as one 16-frame workgroup per CU) and
// has NW = 4 waves, one per SIMD, which split the HIDDEN dimension of a net: first
// the two band classifiers side by side (waves 0,1: left context, waves 2,3: right
// context -- they have the same shape and do not depend on each other), then the
// merger on all four.  Both products are computed transposed so that no data has
// to change lanes between them:
//   layer 1:  S^T[16 hidden x 16 frames] = W1[16 x K] . X^T          A = W1 fragment (HBM/L2, pre-packed)
//                                                                    B = X fragment  (LDS)
//   layer 2:  O^T[16 out x 16 frames]   += W2[16 x 16] . sig(S^T)     A = W2 fragment (HBM/L2, pre-packed)
//                                                                    B = the layer-1 accumulator itself
// The D layout of v_mfma_f32_16x16x4_f32 (row = 4*(lane>>4)+reg, col = lane&15) is
// exactly its B layout for k-slot (lane>>4) if register `reg` is used for k-step
// `reg`; so sig(S^T) feeds layer 2 from registers.  Hidden activations (M x 1500
// floats per net) therefore never exist in LDS or HBM.  Weights are streamed
// from L2/Infinity Cache as whole 1-KiB wave loads (host pre-packs them in
// fragment order) through a ring of registers (mlp_dev.h: RingLoop).  At the end
// of a net its waves' partial O^T tiles meet in LDS (two per net are added while the
// softmax reads them; four are first folded to two), softmax runs on all threads,
// and the result becomes the merger's B image (band nets) or is stored to HBM as
// whole contiguous rows (merger).
//
// Arithmetic contract (tests/: <= 1e-4 max-abs per frame vs the reference):
//   * exp is the reference's FEXP bit trick, bit-emulated (fexp.h:14-21), incl.
//     x86's out-of-range cvttsd2si result: FEXP's f64 product and integer part are
//     exact; the sigmoid's 1/(1+e) is v_rcp_f32 (1 ulp) where the reference divides
//     in f64 and rounds once (fexp.h:33-38; mlp_dev.h SigTile says why);
//   * products accumulate from the bias in ascending k (MFMA = f32 fma chain);
//     layer 2 is split over waves, then folded in a fixed order: deterministic;
//   * projection and normalisation use unfused f32 mul/add in the reference's
//     order (file is compiled with -ffp-contract=off).
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdlib>

#include "lcrc_dev.h"
#include "mlp_dev.h"

namespace phnrec {

// ln() of a band posterior on its way into the merger (sLn, dspc.h:155-160): the device library's logf (1 ulp; a
// v_log_f32 form was measured at -0.4 % and not adopted -- 2 ulp, and a DENORMAL posterior becomes -inf where the
// reference's logf gives -90...-103: profiles/r04_ab_runs.txt 9)
__device__ __forceinline__ float band_ln(float x) { return logf(x); }

// The last-arriver seam of the split-hidden path (guide recipe, write-through form: the partial tiles are stored
// sc1 -- straight through the XCD's L2, so no release fence (an L2 write-back, ~6 us under load) is needed --,
// every wave drains its stores, barrier, ONE lane draws a ticket with a relaxed agent-scope add; the workgroup that
// draws the last ticket acquires at agent scope and reads every slab with plain loads).  Returns true in the
// workgroup that finishes the tile.  Correct for any placement of a tile's workgroups over CUs / XCDs; nobody waits
// for anybody.
typedef unsigned int u4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store_through(f4 v, __amdgpu_buffer_rsrc_t rsrc, int index)
{
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, v), rsrc, index * 16, 0, 16);    // aux 16 = sc1
}

__device__ __forceinline__ bool split_arrive(unsigned *counter, int n_arrivals, int *lds_ticket, int tid)
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        *lds_ticket = (int)__hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if (*lds_ticket != n_arrivals - 1) return false;
    if (tid == 0) {
        __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    return true;
}

// dst[i] = src[i] + src[stride + i] + ... (n_parts slabs, added in slab order: a fixed association)
__device__ __forceinline__ f4 split_sum(const f4 *src, size_t stride, int n_parts)
{
    f4 a = src[0];
    int s = 1;
    for (; s + 3 < n_parts; s += 4) {           // four independent loads in flight, added in order
        const f4 b0 = src[(size_t)s * stride], b1 = src[(size_t)(s + 1) * stride];
        const f4 b2 = src[(size_t)(s + 2) * stride], b3 = src[(size_t)(s + 3) * stride];
        a += b0; a += b1; a += b2; a += b3;
    }
    for (; s < n_parts; s++) a += src[(size_t)s * stride];
    return a;
}

// SPLIT = false: the fused kernel, one workgroup per frame tile, everything up to the posteriors.
// SPLIT = true:  band phase of the split-hidden path (small launches): every frame tile has 2 * p.split_b workgroups,
//                p.split_b per BAND NET; each runs stages 0/1 for its net only and its slice of that net's hidden
//                tiles on all four waves.  With one slice per net (the 1025-2048-frame launches) the workgroup owns
//                the whole net: softmax, ln() and the merger's normalisation follow at once and there is no seam at
//                all; with more slices the last arriver of a (tile, net) adds the partial output tiles first.  The
//                normalised band outputs go to p.gimg ([tile][net][16 frames][16 * n_ot]), from which
//                lcrc_split_merger_kernel builds its operand image.
// PROBES = true: the diagnostic instantiation behind lcrc_posteriors_probe (stage outputs to global memory);
//                the production kernels carry no trace of it (as run-time branches the probe stores cost ~30 % of
//                the band nets' epilogue: their 64-bit address arithmetic was executed for every value).
// Variants whose hidden loops are short (EN: 500 hidden units = 8 / 16 tiles per wave) request the first weight fragments
// of a loop early (RingLoop::begin): the exposed L2 round trip at a loop's start is 3 % of their 16-frame workgroups.
// For the 1500-unit systems it is 0.6 %, less than what the longer register lifetimes cost (r01 A/B run 4).
constexpr bool lcrc_early_requests(int ks1, int ksm, int n_ot, bool exact, bool split)
{
    return exact && !split && ks1 == 64 && ksm == 60 && n_ot == 8;
}

// ARITH = 1: split-f16 arithmetic (mlp_dev.h HalfLoop) -- the operand images hold (high, low) f16 pairs in 32-deep k-steps;
//            an image's size is counted in 1-KiB units per frame tile like the f32 images' k-groups: nkq = 2 * k-steps.
template <int KS1, int KSM, int NOT, int NW, bool EXACT, int FT, bool SPLIT, bool PROBES = false, int ARITH = 0>
__global__ __launch_bounds__(NW * 64) void lcrc_fused_kernel(const LcrcParams p)
{
    static_assert(ARITH == 0 || (EXACT && !SPLIT && !PROBES), "split-f16 arithmetic: the fused kernel of the shipped shapes");
    constexpr int NS1 = (4 * KS1 + 31) / 32, NSM = (4 * KSM + 31) / 32;     // 32-deep k-steps (ARITH = 1)
    constexpr int BM = 16 * FT;                 // frames per workgroup
    constexpr int kTileRows = BM + 2 * kShift;  // mel rows staged
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NT = NW * 64;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nb = p.nbanks;
    // k-groups per frame tile of the operand images.  Run-time shapes use their CLASS's counts as well (groups past the
    // net's own hold zeros): image addresses are then compile-time offsets in the hidden loops (mlp_dev.h RingLoop);
    // only the staging of the normalisation vectors knows the net's own sizes (nkq1_net, nkqm_net).
    const int nkq1 = ARITH ? 2 * NS1 : (KS1 + 3) / 4;
    const int nkqm = ARITH ? 2 * NSM : (KSM + 3) / 4;
    const int nkq1_net = EXACT ? nkq1 : p.net[0].nkq, nkqm_net = EXACT ? nkqm : p.net[2].nkq;
    const int n_ot = EXACT ? NOT : p.n_ot_slab;
    const LdsPlan lp = lcrc_lds_plan(FT, nb, nkq1, nkqm, n_ot);

    float *melT = reinterpret_cast<float *>(smem + lp.mel);
    int *rowlo = reinterpret_cast<int *>(smem + lp.rowinfo);
    int *rowhi = rowlo + BM;
    float *costab = reinterpret_cast<float *>(smem + lp.tabs);
    float *win = costab + 10 * 16;
    // normalisation vectors of the three nets: [mean | dev] per net
    float *nrm_band = reinterpret_cast<float *>(smem + lp.norms);          // [2 nets][2][16*nkq1]
    float *nrm_merger = nrm_band + 4 * 16 * nkq1;                           // [2][16*nkqm]
    float *xf = reinterpret_cast<float *>(smem + lp.xf);
    float *gf = reinterpret_cast<float *>(smem + lp.gf);
    f4 *slab = reinterpret_cast<f4 *>(smem + lp.slab);

    const int split = SPLIT ? p.split_b : 1;                   // slices per band net
    const int tile = SPLIT ? (int)blockIdx.x / (2 * split) : (int)blockIdx.x;
    const int snet = SPLIT ? ((int)blockIdx.x - tile * 2 * split) / split : 0;       // this workgroup's band net
    const int sp = SPLIT ? (int)blockIdx.x - (tile * 2 + snet) * split : 0;
    const int r0 = p.row_first + tile * BM;
    const int tbase = r0 - kShift;

    constexpr bool EARLY = ARITH == 0 && lcrc_early_requests(KS1, KSM, NOT, EXACT, SPLIT);
    // ... and keep the input images in registers as far as they fit: all of them with 16-frame tiles (64 + 60 registers),
    // the first 8 k-groups with 32-frame tiles (hipcc does it by itself for the 1500-unit systems' band nets)
    constexpr int BKQ1 = EARLY ? (FT == 1 ? (KS1 + 3) / 4 : 8) : 0, BKQM = EARLY ? (FT == 1 ? (KSM + 3) / 4 : 8) : 0;
    // the waves' loops of the band pair (waves 0,1: net 0; 2,3: net 1) and of the merger, for the early requests
    RingLoop<KS1, NOT, FT, EXACT, BKQ1> band_loop;
    RingLoop<KSM, NOT, FT, EXACT, BKQM> merger_loop;
    if constexpr (EARLY) {
        const int grp = wave / 2, wig = wave % 2;
        const NetDev &nd = p.net[grp];
        band_loop.setup(nd, reinterpret_cast<const f4 *>(xf) + (size_t)grp * (FT * nkq1 * 64), lane);
        band_loop.begin(wig * ((nd.nht + 1) / 2));
    }

    LCRC_STAMP(p, wave, lane, 0);
#ifdef LCRC_STAMPS
    if (p.stamps && lane == 0) {     // where this wave runs: XCC_ID << 32 | HW_ID (SE, CU, SIMD bits)
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        p.stamps[((size_t)blockIdx.x * 8 + wave) * 16 + 14] = ((unsigned long long)xcc << 32) | hw;
    }
#endif
    // ---- stage 0: utterance bounds per frame, mel tile, tables, zeroed operand images ----
    if (tid < BM) {
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
    {
        // Every global value a thread stages is REQUESTED first (unconditional loads, clamped indices, static
        // trip counts), the operand images are zeroed while the requests travel, then the values are stored:
        // one L2 round trip for the stage instead of one per loop.
        constexpr int kMaxBanks = 23, kMaxW1 = 16 * 16, kMaxWm = 16 * 26;      // the variants' upper bounds
        constexpr int MPT = (kTileRows * kMaxBanks + NT - 1) / NT;             // mel values per thread
        constexpr int WPT = (kMaxW1 + NT - 1) / NT, GPT = (kMaxWm + NT - 1) / NT;
        const int w1n = 16 * nkq1, wmn = 16 * nkqm, ntile = kTileRows * nb;          // LDS layout
        const int w1v = 16 * nkq1_net, wmv = 16 * nkqm_net;                            // what the nets' arrays hold
        const int nmel = p.n_rows * nb;          // (rows * banks < 2^31: the API bounds a call's frames)
        float mv[MPT], tv, wv[WPT][4], gv[GPT][2];
#pragma unroll
        for (int q = 0; q < MPT; q++) {
            // 32-bit offsets from the scalar base: `global_load v, v_off, s[mel]` instead of 64-bit per-lane addresses
            const int g = tbase * nb + tid + q * NT;
            mv[q] = p.mel[(unsigned)max(0, min(nmel - 1, g))];
        }
        static_assert((KS1 + 3) / 4 <= 16 && (KSM + 3) / 4 <= 26, "staging bounds");
        const float *tsrc = tid < 160 ? p.costab + tid : p.win + (min(tid, 191) - 160);
        tv = *tsrc;
#pragma unroll
        for (int q = 0; q < WPT; q++) {
            const unsigned i = (unsigned)min(tid + q * NT, w1v - 1);     // unsigned: 32-bit offset from a scalar base
            wv[q][0] = p.net[0].mean[i]; wv[q][1] = p.net[0].dev[i];
            wv[q][2] = p.net[1].mean[i]; wv[q][3] = p.net[1].dev[i];
        }
#pragma unroll
        for (int q = 0; q < GPT; q++) {
            const unsigned i = (unsigned)min(tid + q * NT, wmv - 1);
            gv[q][0] = p.net[2].mean[i]; gv[q][1] = p.net[2].dev[i];
        }
        LCRC_FENCE();
        const f4 zero = {0.f, 0.f, 0.f, 0.f};             // pads of the operand images must be zeros
        f4 *zx = reinterpret_cast<f4 *>(xf), *zg = reinterpret_cast<f4 *>(gf);
        const int nx = 2 * FT * nkq1 * 64, ng = FT * nkqm * 64;
        for (int i = tid; i < nx; i += NT) zx[i] = zero;
        for (int i = tid; i < ng; i += NT) zg[i] = zero;
        LCRC_FENCE();
#pragma unroll
        for (int q = 0; q < MPT; q++) {
            const int i = tid + q * NT;
            const int row = tbase + i / nb;
            if (i < ntile) melT[i] = (row >= 0 && row < p.n_rows) ? mv[q] : 0.0f;
        }
        if (tid < 10 * 16 + 2 * 16) costab[tid] = tv;
#pragma unroll
        for (int q = 0; q < WPT; q++) {
            const int i = tid + q * NT;
            if (i < w1v) {
                nrm_band[i] = wv[q][0]; nrm_band[w1n + i] = wv[q][1];
                nrm_band[2 * w1n + i] = wv[q][2]; nrm_band[3 * w1n + i] = wv[q][3];
            }
        }
#pragma unroll
        for (int q = 0; q < GPT; q++) {
            const int i = tid + q * NT;
            if (i < wmv) { nrm_merger[i] = gv[q][0]; nrm_merger[wmn + i] = gv[q][1]; }
        }
    }
    __syncthreads();

    LCRC_STAMP(p, wave, lane, 1);
    // ---- stage 1: window * DCT projection + input normalisation (traps.cpp:285-343,
    //      dspc.h:107-112,206-233, nn.cpp:702-716) as small MFMA products:
    //      out[frame][c] = sum_tap (x[frame][tap] * win[tap]) * D[tap][c],  D[:,0] = 1 (C0), D[:,1..10] = cos.
    //      A = windowed half context (16 frames x 4 taps per step, gathered from the mel tile with
    //      clamped indices), B = basis (4 registers per lane, loaded once).  Taps are summed in
    //      ascending order as in the reference; the MFMA fuses each multiply-add (<= 1 ulp apart
    //      from the reference's separate mul and add). ----
    {
        const int K = p.net[0].n_inp;            // nbanks * 11
        const float normc = p.normc;
        const int n_rows = p.n_rows, row_end = p.row_end;
        float *const dbg_in0 = p.dbg_in0, *const dbg_in1 = p.dbg_in1;    // (probes: whole-range launches only)
        const int g = lane >> 4, c = lane & 15;
        float basis[4];
#pragma unroll
        for (int s4 = 0; s4 < 4; s4++) {
            const int tap = 4 * s4 + g;
            basis[s4] = c == 0 ? 1.0f : (c < kNCoef ? costab[(c - 1) * 16 + tap] : 0.0f);
        }
        // per-lane constants of both frame tiles: clamp bounds and centre row
        int rr[FT], lo[FT], hi[FT];
#pragma unroll
        for (int f = 0; f < FT; f++) {
            const int i = 16 * f + c;            // this lane's frame as an A-operand row
            rr[f] = min(r0 + i, n_rows - 1) - kShift;
            lo[f] = rowlo[i];
            hi[f] = rowhi[i];
        }
        const int cc = min(c, kNCoef - 1);       // loads below stay unconditional (clamped index)
        // The clamped source rows of a lane's taps depend on the half context (n) only, not on the band:
        // their tile offsets and the window values are formed once, an item adds its band.
        // NN half contexts per workgroup: both (fused kernel), or the one of this workgroup's band net (split path);
        // slot q of the arrays below is half context nq = n_base + q.
        constexpr int NN = SPLIT ? 1 : 2;
        const int n_base = SPLIT ? snet : 0;
        int roff[NN][FT][4];
        float wv[NN][4];
#pragma unroll
        for (int q = 0; q < NN; q++)
#pragma unroll
            for (int s4 = 0; s4 < 4; s4++) {
                const int tap = 4 * s4 + g, n = n_base + q;
                wv[q][s4] = win[n * kHalf + tap];
#pragma unroll
                for (int f = 0; f < FT; f++) {
                    const int srow = max(lo[f], min(hi[f], rr[f] + n * kShift + tap));
                    roff[q][f][s4] = (srow - tbase) * nb;
                }
            }
        // Bands are dealt to the waves; an item is one band's half contexts together (NN * FT independent MFMA chains).
        // The operand values and the normalisation constants of band b + NW are requested (independent LDS reads) before
        // the MFMAs of band b: the compiler cannot move LDS reads above the previous item's operand-image stores by itself.
        {
            // (offsets, not pointers: through an array of pointers hipcc loses the LDS address space)
            int mofs[NN];
#pragma unroll
            for (int q = 0; q < NN; q++) mofs[q] = (n_base + q) * 32 * nkq1;
            auto gather = [&](int b, float (&x)[NN][FT][4], float (&mk)[NN], float (&dk)[NN]) {
                const int bc = min(b, nb - 1);
#pragma unroll
                for (int q = 0; q < NN; q++) {
#pragma unroll
                    for (int s4 = 0; s4 < 4; s4++)
#pragma unroll
                        for (int f = 0; f < FT; f++) x[q][f][s4] = melT[roff[q][f][s4] + bc] * wv[q][s4];
                    mk[q] = nrm_band[mofs[q] + bc * kNCoef + cc];
                    dk[q] = nrm_band[mofs[q] + 16 * nkq1 + bc * kNCoef + cc];
                }
            };
            float xw[NN][FT][4], xn[NN][FT][4], mk[NN], dk[NN], mkn[NN], dkn[NN];
            gather(wave, xw, mk, dk);
            for (int b = wave; b < nb; b += NW) {
                gather(b + NW, xn, mkn, dkn);
                const int k = b * kNCoef + cc;
                f4 acc[NN][FT];
#pragma unroll
                for (int q = 0; q < NN; q++)
#pragma unroll
                    for (int f = 0; f < FT; f++) acc[q][f] = (f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int s4 = 0; s4 < 4; s4++)
#pragma unroll
                    for (int q = 0; q < NN; q++)
#pragma unroll
                        for (int f = 0; f < FT; f++) acc[q][f] = mfma16x16x4(xw[q][f][s4], basis[s4], acc[q][f]);
                if (c < kNCoef) {                    // D layout: row = frame 16f + 4g + reg, col = c
                    // B-image address of (frame fr, input k): see xf_store; only `fr` varies below
                    const int kbase = (((k >> 4) * 64) + 16 * (k & 3)) * 4 + ((k >> 2) & 3);
#pragma unroll
                    for (int q = 0; q < NN; q++) {
                        const int n = n_base + q;
                        float *img = xf + (size_t)n * (FT * nkq1 * 256);
#pragma unroll
                        for (int f = 0; f < FT; f++) {
#pragma unroll
                            for (int reg = 0; reg < 4; reg++) {
                                const float val = acc[q][f][reg] * normc;        // CalcC0 / sDCT scaling
                                float v = val - mk[q];                           // Normalize nn.cpp:702-716
                                v *= dk[q];
                                if constexpr (ARITH == 1)
                                    h2_img_store(img, f * (2 * NS1 * 1024) + (4 * g + reg) * 16 + h2_k_ofs(k), NS1 * 1024, v);
                                else img[f * nkq1 * 256 + (4 * g + reg) * 4 + kbase] = v;
                            }
                        }
                        float *dbg = PROBES ? (n == 0 ? dbg_in0 : dbg_in1) : nullptr;
                        if (PROBES && dbg) {             // stage probe (diagnostic instantiation only)
#pragma unroll
                            for (int f = 0; f < FT; f++)
#pragma unroll
                                for (int reg = 0; reg < 4; reg++) {
                                    const int fr = 16 * f + 4 * g + reg;
                                    if (r0 + fr < row_end) dbg[(size_t)(r0 + fr) * K + k] = acc[q][f][reg] * normc;
                                }
                        }
                    }
                }
#pragma unroll
                for (int q = 0; q < NN; q++) {
#pragma unroll
                    for (int f = 0; f < FT; f++)
#pragma unroll
                        for (int s4 = 0; s4 < 4; s4++) xw[q][f][s4] = xn[q][f][s4];
                    mk[q] = mkn[q];
                    dk[q] = dkn[q];
                }
            }
        }
    }
    __syncthreads();

    LCRC_STAMP(p, wave, lane, 10);              // projection done
    // ---- stage 2: the two band nets SIDE BY SIDE (waves 0,1: left context, waves 2,3: right context);
    //      ln() of their outputs, normalised for the merger, goes straight from the softmax registers
    //      into the merger's operand image ----
    const NetDev &nm = p.net[2];
    const float *mmean = nrm_merger, *mdev = nrm_merger + 16 * nkqm;
    {
        // kernel arguments used per value live in locals: fields of `p` are re-read from the kernarg
        // segment (a scalar load + wait) behind every memory fence and barrier
        const int O0 = p.net[0].n_out, O1 = p.net[1].n_out, n_rows = p.row_end, merger_inp = nm.n_inp;
        float *const dbg_p0 = p.dbg_p0, *const dbg_p1 = p.dbg_p1, *const dbg_g = p.dbg_g;
        const bool probes = PROBES && (dbg_p0 || dbg_p1 || dbg_g);  // diagnostic instantiation only
        // Row epilogue: a lane's NV band posteriors of frame i (outputs o = part + LPF*j of net n) become
        // merger inputs k = n*O0 + o.  Straight-line and batched: all ln() first, then the normalisation
        // constants of all values requested from LDS at once, then the (predicated) stores into the operand
        // image.  (One basic block per value -- the obvious form -- exposed an LDS round trip per value: 14 K
        // cycles per 32-frame tile against 12.4 K, profiles/r02_ab_runs.txt item 3.)
        auto epi = [&](int n, int i, int part, auto lpf, const auto &q, int O) {
            constexpr int LPF = decltype(lpf)::value;
            constexpr int NV = sizeof(q) / sizeof(float);
            const int kofs = n * O0;
            const int kb = kofs + part;                              // k of value j = kb + LPF*j
            float gl[NV], mk[NV], dk[NV];
#pragma unroll
            for (int j = 0; j < NV; j++) {                           // (reads past n_inp stay inside the padded arrays)
                mk[j] = mmean[kb + LPF * j];
                dk[j] = mdev[kb + LPF * j];
            }
#pragma unroll
            for (int j = 0; j < NV; j++) gl[j] = q[j] > 0.0f ? band_ln(q[j]) : 0.0f;     // sLn dspc.h:155-160
            if (PROBES && probes && r0 + i < n_rows) {
                float *dp = n == 0 ? dbg_p0 : dbg_p1;
#pragma unroll
                for (int j = 0; j < NV; j++) {
                    const int o = part + LPF * j;
                    if (o < O && dp) dp[(size_t)(r0 + i) * (n == 0 ? O0 : O1) + o] = q[j];
                    if (o < O && dbg_g) dbg_g[(size_t)(r0 + i) * merger_inp + kofs + o] = gl[j];
                }
            }
            // B-image address of (frame i, input k), see xf_store: LPF is a multiple of 4, so k & 3 is the
            // lane's own constant and only (k >> 2) moves with j
            float *const img = gf + ((i >> 4) * nkqm * 64 + (i & 15) + 16 * (kb & 3)) * 4;
            const int t0 = kb >> 2;
            // split-f16 image: value j lies (LPF / 8) * 256 bytes behind value j - 1 (LPF = 4: behind value j - 2)
            const int hf = (i >> 4) * (2 * NSM * 1024) + (i & 15) * 16;
            const int h0 = hf + h2_k_ofs(kb), h1 = hf + h2_k_ofs(kb + 4);
#pragma unroll
            for (int j = 0; j < NV; j++) {
                float v = gl[j] - mk[j];                             // Normalize nn.cpp:702-716
                v *= dk[j];
                const int t = t0 + (LPF / 4) * j;
                // exact variants: outputs below 16 * (NOT - 1) are valid whatever n_out is -- no compare, no branch
                if ((EXACT && LPF * j + LPF <= 16 * (NOT - 1)) || part + LPF * j < O) {
                    if constexpr (ARITH == 1)
                        h2_img_store(gf, LPF >= 8 ? h0 + j * (LPF / 8) * 256 : ((j & 1) ? h1 : h0) + (j >> 1) * 256, NSM * 1024, v);
                    else img[(t >> 2) * 256 + (t & 3)] = v;
                }
            }
        };
        if constexpr (SPLIT) {
            // ---- band phase of the split-hidden path: slice `sp` of band net `snet`'s hidden tiles, on all four waves ----
            constexpr int NTH = NW * 64;
            constexpr int OV = EXACT ? 16 * (NOT - 1) : 0;
            const NetDev &nd = p.net[snet];
            const int hbeg = min(nd.nht, sp * p.tps_b), hend = min(nd.nht, hbeg + p.tps_b);
            const int tpw = (hend - hbeg + NW - 1) / NW;
            const int ht0 = hbeg + wave * tpw;
            const int slab_f4 = FT * n_ot * 64;
            f4 *const slab23 = reinterpret_cast<f4 *>(smem + lp.slab23);
            {
                f4 acc[NOT][FT];
                hidden_range<KS1, NOT, EXACT, FT>(nd, reinterpret_cast<const f4 *>(xf) + (size_t)snet * (FT * nkq1 * 64), min(hend, ht0),
                                                  min(hend, ht0 + tpw), sp == 0 && wave == 0, lane, acc);
                __syncthreads();                                 // slab23 lies over the operand images
                store_partial<NOT, EXACT, FT>((wave < 2 ? slab : slab23) + (wave & 1) * slab_f4, EXACT ? NOT : nd.n_ot, lane, acc);
                __syncthreads();
            }
            // Row epilogue: a lane's band posteriors of frame i -> ln(), the merger's normalisation (its mean / dev of
            // inputs snet * O0 + o) -> the (tile, net) block of p.gimg, rows of 16 * n_ot floats
            const int OP = 16 * n_ot;
            float *const blk = reinterpret_cast<float *>(p.gimg) + (size_t)(tile * 2 + snet) * (BM * OP);
            auto epi_split = [&](int, int i, int part, auto lpf, const auto &q, int O) {
                constexpr int LPF = decltype(lpf)::value;
                constexpr int NV = sizeof(q) / sizeof(float);
                const int kb = snet * O0 + part;
                float gl[NV], mk[NV], dk[NV];
#pragma unroll
                for (int j = 0; j < NV; j++) {                       // (reads past n_inp stay inside the padded arrays)
                    mk[j] = mmean[kb + LPF * j];
                    dk[j] = mdev[kb + LPF * j];
                }
#pragma unroll
                for (int j = 0; j < NV; j++) gl[j] = q[j] > 0.0f ? band_ln(q[j]) : 0.0f;     // sLn dspc.h:155-160
                float *const row = blk + i * OP + part;
#pragma unroll
                for (int j = 0; j < NV; j++) {
                    float v = gl[j] - mk[j];                         // Normalize nn.cpp:702-716
                    v *= dk[j];
                    if ((EXACT && LPF * j + LPF <= OV) || part + LPF * j < O) row[LPF * j] = v;
                }
            };
            const float *s01 = reinterpret_cast<const float *>(slab), *s23 = reinterpret_cast<const float *>(slab23);
            if (split == 1) {
                // the whole net is here: the four waves' partial tiles are added while the softmax reads them -- no seam
                softmax_rows<NOT, NW, FT, 1, 4, OV>(p, &nd, s01, s01 + slab_f4 * 4, s23, s23 + slab_f4 * 4, s01, s01, lane, wave, epi_split);
                return;
            }
            // the workgroup's partial tile ((wave 0 + 1) + (wave 2 + 3)) leaves as whole 1-KiB write-through wave stores
            const __amdgpu_buffer_rsrc_t mine = __builtin_amdgcn_make_buffer_rsrc(
                reinterpret_cast<f4 *>(p.part) + (size_t)blockIdx.x * slab_f4, 0, slab_f4 * 16, 0x00020000);
            for (int i = tid; i < slab_f4; i += NTH)
                store_through((slab[i] + slab[slab_f4 + i]) + (slab23[i] + slab23[slab_f4 + i]), mine, i);
            int *const ticket = reinterpret_cast<int *>(smem + lp.total);
            if (!split_arrive(p.cnt + 4 * tile + snet, split, ticket, tid)) return;
            // last arriver of (tile, net): add the slices' partials in slice order, then softmax + ln()
            const f4 *const first = reinterpret_cast<const f4 *>(p.part) + (size_t)(tile * 2 + snet) * split * slab_f4;
            for (int i = tid; i < slab_f4; i += NTH) slab[i] = split_sum(first + i, (size_t)slab_f4, split);
            __syncthreads();
            softmax_rows<NOT, NW, FT, 1, 1, OV>(p, &nd, s01, s01, s01, s01, s01, s01, lane, wave, epi_split);
            return;
        } else {
        // (the sequential alternative -- one band net after the other on four waves -- was 1.6-3 % slower in
        //  same-GPU A/B runs, profiles/r01_ab_runs.txt)
        if constexpr (EARLY) {
            auto begin_merger = [&]() {
                merger_loop.setup(nm, reinterpret_cast<const f4 *>(gf), lane);
                merger_loop.begin(wave * ((nm.nht + NW - 1) / NW));
            };
            run_net<KS1, NOT, NW, EXACT, FT, 2, true, BKQ1>(p, 2, p.net, reinterpret_cast<const f4 *>(xf), FT * nkq1 * 64, slab,
                                                      reinterpret_cast<f4 *>(smem + lp.slab23), n_ot, lane, wave, epi,
                                                      &band_loop, begin_merger);
        } else {
            run_net<KS1, NOT, NW, EXACT, FT, 2, false, 0, ARITH>(p, 2, p.net, reinterpret_cast<const f4 *>(xf), FT * nkq1 * 64, slab,
                                                                 reinterpret_cast<f4 *>(smem + lp.slab23), n_ot, lane, wave, epi);
        }
        LCRC_STAMP(p, wave, lane, 3);           // softmax + ln() done
        }
    }

    if constexpr (!SPLIT) {
    // ---- stage 3: merger; posteriors are gathered as contiguous rows in LDS (slab 0 is
    //      free again) and leave as one linear, 16-byte-per-lane copy ----
    {
        const int O = nm.n_out;
        float *outbuf = reinterpret_cast<float *>(slab);
        // the writer path's settings in locals (kernarg fields are re-read behind every fence)
        WriterPathEpilogue epi;
        epi.outbuf = outbuf; epi.O = O; epi.f0 = p.out_func[0]; epi.f1 = p.out_func[1]; epi.be = p.out_be;
        epi.ovalid = EXACT ? 16 * (NOT - 1) : 0;
        for (int i = 0; i < 3; i++) { epi.c0[i] = p.out_c[0][i]; epi.c1[i] = p.out_c[1][i]; }
        for (int i = 0; i < 2; i++) { epi.l0[i] = p.out_l[0][i]; epi.l1[i] = p.out_l[1][i]; }
        if constexpr (EARLY)
            run_net<KSM, NOT, NW, EXACT, FT, 1, true, BKQM>(p, 8, &nm, reinterpret_cast<const f4 *>(gf), 0, slab,
                                                      reinterpret_cast<f4 *>(smem + lp.slab23), n_ot, lane, wave, epi, &merger_loop);
        else
            run_net<KSM, NOT, NW, EXACT, FT, 1, false, 0, ARITH>(p, 8, &nm, reinterpret_cast<const f4 *>(gf), 0, slab,
                                                                 reinterpret_cast<f4 *>(smem + lp.slab23), n_ot, lane, wave, epi);
        LCRC_STAMP(p, wave, lane, 9);
        const int rows = min(BM, p.row_end - r0);
        const int total = rows * O;
        float *dst = p.post + (size_t)(r0 - p.row_first) * O;      // 16*O*4 bytes per frame tile: 16-byte aligned
        const int n4 = total >> 2;
        for (int i = tid; i < n4; i += NT)
            reinterpret_cast<f4 *>(dst)[i] = reinterpret_cast<const f4 *>(outbuf)[i];
        for (int i = (n4 << 2) + tid; i < total; i += NT) dst[i] = outbuf[i];
    }
    }
    LCRC_STAMP(p, wave, lane, 11);
}


// Merger phase of the split-hidden path: p.split_m workgroups per frame tile, each on tps_m hidden tiles of the
// merger (its four waves share them); the last arriver adds the partial tiles in slice order, runs the
// softmax and the writer path and stores the tile's posteriors.
template <int KSM, int NOT, int NW, bool EXACT, int FT>
__global__ __launch_bounds__(NW * 64) void lcrc_split_merger_kernel(const LcrcParams p)
{
    constexpr int BM = 16 * FT, NT = NW * 64;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const NetDev &nm = p.net[2];
    constexpr int nkqm = (KSM + 3) / 4;                         // the class's k-groups (see lcrc_fused_kernel)
    const int n_ot = EXACT ? NOT : p.n_ot_slab;
    const int slab_f4 = FT * n_ot * 64;
    f4 *const gf = reinterpret_cast<f4 *>(smem);
    f4 *const slab = gf + FT * nkqm * 64;                       // four wave slabs
    int *const ticket = reinterpret_cast<int *>(slab + 4 * slab_f4);
    const int split = p.split_m;
    const int tile = (int)blockIdx.x / split, sp = (int)blockIdx.x - tile * split;
    const int r0 = p.row_first + tile * BM;

    {
        // operand image from the band phase's outputs ([tile][net][16 frames][16 * n_ot_band], already ln()ed and
        // normalised): zero pads first, then 16 threads per frame scatter that frame's values (merger input k = net * O0 + o)
        const f4 zero = {0.f, 0.f, 0.f, 0.f};
        for (int i = tid; i < FT * nkqm * 64; i += NT) gf[i] = zero;
        __syncthreads();
        const int O0 = p.net[0].n_out, O1 = p.net[1].n_out, OP = 16 * n_ot;
        const float *const blk = reinterpret_cast<const float *>(p.gimg) + (size_t)tile * 2 * (BM * OP);
        float *const img = reinterpret_cast<float *>(gf);
        for (int fr = tid >> 4; fr < BM; fr += NT / 16) {
            const int u = tid & 15;
            for (int o = u; o < O0; o += 16) xf_store(img, nkqm, fr, o, blk[fr * OP + o]);
            for (int o = u; o < O1; o += 16) xf_store(img, nkqm, fr, O0 + o, blk[(BM + fr) * OP + o]);
        }
    }
    __syncthreads();
    {
        const int hbeg = min(nm.nht, sp * p.tps_m), hend = min(nm.nht, hbeg + p.tps_m);
        const int tpw = (hend - hbeg + NW - 1) / NW;
        const int ht0 = hbeg + wave * tpw;
        f4 acc[NOT][FT];
        hidden_range<KSM, NOT, EXACT, FT>(nm, gf, min(hend, ht0), min(hend, ht0 + tpw), sp == 0 && wave == 0, lane, acc);
        store_partial<NOT, EXACT, FT>(slab + wave * slab_f4, EXACT ? NOT : nm.n_ot, lane, acc);
        __syncthreads();
    }
    const __amdgpu_buffer_rsrc_t mine = __builtin_amdgcn_make_buffer_rsrc(
        reinterpret_cast<f4 *>(p.part) + (size_t)blockIdx.x * slab_f4, 0, slab_f4 * 16, 0x00020000);
    for (int i = tid; i < slab_f4; i += NT)
        store_through((slab[i] + slab[slab_f4 + i]) + (slab[2 * slab_f4 + i] + slab[3 * slab_f4 + i]), mine, i);
    if (!split_arrive(p.cnt + 4 * tile + 2, split, ticket, tid)) return;
    const f4 *const first = reinterpret_cast<const f4 *>(p.part) + (size_t)tile * split * slab_f4;
    for (int i = tid; i < slab_f4; i += NT) slab[i] = split_sum(first + i, (size_t)slab_f4, split);
    __syncthreads();

    const int O = nm.n_out;
    float *outbuf = reinterpret_cast<float *>(slab + slab_f4);   // slabs 1.. are free
    WriterPathEpilogue epi;
    epi.outbuf = outbuf; epi.O = O; epi.f0 = p.out_func[0]; epi.f1 = p.out_func[1]; epi.be = p.out_be;
    epi.ovalid = EXACT ? 16 * (NOT - 1) : 0;
    for (int i = 0; i < 3; i++) { epi.c0[i] = p.out_c[0][i]; epi.c1[i] = p.out_c[1][i]; }
    for (int i = 0; i < 2; i++) { epi.l0[i] = p.out_l[0][i]; epi.l1[i] = p.out_l[1][i]; }
    const float *s0 = reinterpret_cast<const float *>(slab);
    softmax_rows<NOT, NW, FT, 1, 1, (EXACT ? 16 * (NOT - 1) : 0)>(p, &nm, s0, s0, s0, s0, s0, s0, lane, wave, epi);
    __syncthreads();
    const int rows = min(BM, p.row_end - r0);
    const int total = rows * O;
    float *dst = p.post + (size_t)(r0 - p.row_first) * O;
    const int n4 = total >> 2;
    for (int i = tid; i < n4; i += NT)
        reinterpret_cast<f4 *>(dst)[i] = reinterpret_cast<const f4 *>(outbuf)[i];
    for (int i = (n4 << 2) + tid; i < total; i += NT) dst[i] = outbuf[i];
}

// ---- variants --------------------------------------------------------------------------
// (layer-1 k-steps of the band nets, of the merger, output tiles) of the shipped systems:
// CZ 165->1500->138 / 276  HU ..->186 / 372  RU 165->1400->159 / 318  EN 253->500->120 / 240
namespace {

constexpr int kNW = 4;   // waves per workgroup: one per SIMD (two per SIMD do not pay)
// Run-time-shape ("generic") size classes: every MFMA group of the class runs (zero fragments past the net's own
// sizes), so a class costs its maxima.  Small: <= 16 banks (176 inputs), <= 144 outputs; large: <= 23 banks, <= 208 outputs.
constexpr int kGenSKS1 = 44, kGenSKSM = 72, kGenSNOT = 9;
constexpr int kGenKS1 = 64, kGenKSM = 104, kGenNOT = 13;

struct Variant {
    const char *name;
    int ks1, ksm, n_ot;    // exact: the shape; classes: the maxima
    bool exact;
    const void *fn[2];     // [FT - 1]: 16- and 32-frame workgroups
    const void *split_band, *split_merger;   // split-hidden path, 16-frame tiles
    const void *probe;     // 16-frame workgroups with the stage probes (lcrc_posteriors_probe)
    const void *h2[2];     // split-f16 arithmetic, [FT - 1] (shipped shapes only)
};

#define LCRC_KERNEL(KS1, KSM, NOT, EX) \
    {reinterpret_cast<const void *>(&lcrc_fused_kernel<KS1, KSM, NOT, kNW, EX, 1, false>), \
     reinterpret_cast<const void *>(&lcrc_fused_kernel<KS1, KSM, NOT, kNW, EX, 2, false>)}, \
    reinterpret_cast<const void *>(&lcrc_fused_kernel<KS1, KSM, NOT, kNW, EX, 1, true>), \
    reinterpret_cast<const void *>(&lcrc_split_merger_kernel<KSM, NOT, kNW, EX, 1>), \
    reinterpret_cast<const void *>(&lcrc_fused_kernel<KS1, KSM, NOT, kNW, EX, 1, false, true>)

#define LCRC_KERNEL_H2(KS1, KSM, NOT) \
    {reinterpret_cast<const void *>(&lcrc_fused_kernel<KS1, KSM, NOT, kNW, true, 1, false, false, 1>), \
     reinterpret_cast<const void *>(&lcrc_fused_kernel<KS1, KSM, NOT, kNW, true, 2, false, false, 1>)}

const Variant kVariants[] = {
    {"cz_42_69_9", 42, 69, 9, true, LCRC_KERNEL(42, 69, 9, true), LCRC_KERNEL_H2(42, 69, 9)},
    {"hu_42_93_12", 42, 93, 12, true, LCRC_KERNEL(42, 93, 12, true), LCRC_KERNEL_H2(42, 93, 12)},
    {"ru_42_80_10", 42, 80, 10, true, LCRC_KERNEL(42, 80, 10, true), LCRC_KERNEL_H2(42, 80, 10)},
    {"en_64_60_8", 64, 60, 8, true, LCRC_KERNEL(64, 60, 8, true), LCRC_KERNEL_H2(64, 60, 8)},
    {"generic_44_72_9", kGenSKS1, kGenSKSM, kGenSNOT, false, LCRC_KERNEL(kGenSKS1, kGenSKSM, kGenSNOT, false), {nullptr, nullptr}},
    {"generic_64_104_13", kGenKS1, kGenKSM, kGenNOT, false, LCRC_KERNEL(kGenKS1, kGenKSM, kGenNOT, false), {nullptr, nullptr}},
};
constexpr int kNVariants = sizeof kVariants / sizeof kVariants[0];

const Variant *pick(const NetDev *nets)
{
    if (nets[1].ksteps != nets[0].ksteps) return nullptr;
    const int n_ot = lcrc_n_ot_slab(nets);
    for (const Variant &v : kVariants) {           // exact shapes first, then the smallest class that holds the model
        if (v.exact ? (v.ks1 == nets[0].ksteps && v.ksm == nets[2].ksteps && v.n_ot == nets[0].n_ot &&
                       v.n_ot == nets[1].n_ot && v.n_ot == nets[2].n_ot)
                    : (nets[0].ksteps <= v.ks1 && nets[2].ksteps <= v.ksm && n_ot <= v.n_ot))
            return &v;
    }
    return nullptr;
}

}  // namespace

const char *lcrc_variant_for(const NetDev *nets, int nbanks, unsigned *lds_bytes)
{
    const Variant *v = pick(nets);
    if (!v) return nullptr;
    // 32-frame workgroups when their LDS image fits, else 16-frame ones only (same results)
    // (operand images are laid out with the variant's -- for run-time shapes: the class's -- k-groups)
    const int k1 = (v->ks1 + 3) / 4, km = (v->ksm + 3) / 4;
    unsigned total = lcrc_lds_plan(2, nbanks, k1, km, lcrc_n_ot_slab(nets)).total;
    if (total > 160u * 1024u) total = lcrc_lds_plan(1, nbanks, k1, km, lcrc_n_ot_slab(nets)).total;
    if (lds_bytes) *lds_bytes = total;
    if (total > 160u * 1024u) return nullptr;
    return v->name;
}

bool lcrc_has_split_f16(const NetDev *nets)
{
    const Variant *v = pick(nets);
    return v && v->h2[0];
}

void lcrc_split_scratch(const NetDev *nets, int wgs, size_t *part_bytes, size_t *gimg_bytes, size_t *cnt_bytes)
{
    const size_t slab = (size_t)lcrc_n_ot_slab(nets) * 1024u;          // one 16-frame partial tile
    *part_bytes = (size_t)wgs * slab;                                  // one per workgroup in either phase
    *gimg_bytes = (size_t)wgs * slab;                                  // band outputs: [tile][net][16][16 * n_ot]; tiles <= wgs / 2
    *cnt_bytes = (size_t)wgs * 4 * sizeof(unsigned);                   // [tile][band net 0, band net 1, merger, -]
}

namespace {

// > 64 KiB of dynamic LDS has to be granted per function and per device: once, not per launch
// (atomics: several host threads launch on their own contexts; the worst case is a repeated grant)
hipError_t grant_lds(const void *fn, int vi, int slot, int dev)
{
    static std::atomic<bool> granted[kNVariants][7][64] = {};
    const bool cached = dev >= 0 && dev < 64;
    if (cached && granted[vi][slot][dev]) return hipSuccess;
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e == hipSuccess && cached) granted[vi][slot][dev] = true;
    return e;
}

// Workgroups per frame tile of the split-hidden path for a launch of `tiles` 16-frame tiles, 1 = fused kernel.
// Automatic choice: launches whose 16-frame tiles would leave at least half of the CUs idle spread every
// tile's hidden dimension over as many workgroups as CUs allow, up to kSplitMax (beyond it the last
// arriver's slab reads cost more than the shorter hidden slices save).
constexpr int kSplitMax = 12;
int choose_split(const LcrcParams &p, int tiles, int n_cu)
{
    if (!p.part || !p.gimg || !p.cnt || p.split_hint == 1 || tiles <= 0) return 1;
    if (p.dbg_in0 || p.dbg_in1 || p.dbg_p0 || p.dbg_p1 || p.dbg_g || p.stamps) return 1;   // probes: fused kernel only
    if (p.split_hint > 1) return max(1, min(p.split_hint, p.split_cap_wgs / tiles));   // forced: may oversubscribe the CUs (tuning)
    int s = min(n_cu / tiles, kSplitMax);
    s = min(s, p.split_cap_wgs / tiles);
    // at least one hidden tile per wave in both phases, or the extra workgroups only add to the seam
    s = min(s, max(1, min(max(p.net[0].nht, p.net[1].nht) / 2, p.net[2].nht / 4)));
    if (s < 2) return 1;
    // Worth it?  A tile's hidden loops take about h_us on one CU (its MFMA work at ~80 % of a CU's f32 rate);
    // splitting saves h_us * (1 - 1/s) and pays the two seams (publish, ticket, slab reads by the last arriver, the
    // second launch): ~12 us, idle or full chip, with write-through slab stores (measured,
    // profiles/r02_small_launch_sweep.txt: EN at 2048 frames gains 6 us with two workgroups per tile, CZ 39 us).
    double macs = 0.0;
    for (int i = 0; i < 3; i++) macs += (double)p.net[i].n_hid * (p.net[i].n_inp + p.net[i].n_out);
    const double h_us = 16.0 * 2.0 * macs / 490e3;          // 490 GFLOP/s per CU
    const double seam_us = 13.0;
    return h_us * (1.0 - 1.0 / s) >= seam_us ? s : 1;
}

}  // namespace

hipError_t lcrc_launch(const LcrcParams &p, hipStream_t stream, const char **variant_name)
{
    const Variant *v = pick(p.net);
    if (!v) return hipErrorInvalidValue;
    const int k1 = (v->ks1 + 3) / 4, km = (v->ksm + 3) / 4;     // k-groups of the operand images (the class's for run-time shapes)
    const bool fits32 = lcrc_lds_plan(2, p.nbanks, k1, km, lcrc_n_ot_slab(p.net)).total <= 160u * 1024u;
    if (!fits32 && lcrc_lds_plan(1, p.nbanks, k1, km, lcrc_n_ot_slab(p.net)).total > 160u * 1024u)
        return hipErrorInvalidValue;
    if (variant_name) *variant_name = v->name;
    LcrcParams args = p;
    if (args.row_end == 0) { args.row_first = 0; args.row_end = p.n_rows; }
    if (args.row_first < 0 || args.row_end > p.n_rows || args.row_first > args.row_end) return hipErrorInvalidValue;
    const int rows = args.row_end - args.row_first;
    if (p.n_rows <= 0 || rows <= 0) return hipSuccess;
    static std::atomic<int> cus[64] = {};
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    const bool cached = dev >= 0 && dev < 64;
    int n_cu = cached ? cus[dev].load() : 0;
    if (n_cu == 0) {
        e = hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev);
        if (e != hipSuccess) return e;
        if (cached) cus[dev] = n_cu;
    }
    const int vi = (int)(v - kVariants);
    args.n_ot_slab = lcrc_n_ot_slab(p.net);
    void *kargs[] = {&args};
    const dim3 block(kNW * 64);

    // ---- split-f16 arithmetic (lcrc_set_arithmetic): the fused kernel at every launch size ----
    if (p.arith == 1) {
        const bool probes = p.dbg_in0 || p.dbg_in1 || p.dbg_p0 || p.dbg_p1 || p.dbg_g;
        if (!v->h2[0] || !p.net[0].w1h || !p.net[1].w1h || !p.net[2].w1h || probes) return hipErrorInvalidValue;
        const int nk1 = 2 * ((4 * v->ks1 + 31) / 32), nkm = 2 * ((4 * v->ksm + 31) / 32);
        int ft = p.tile_frames == 16 ? 1 : p.tile_frames == 32 ? 2 : ((rows + 31) / 32 <= n_cu / 2 ? 1 : 2);
        if (lcrc_lds_plan(2, p.nbanks, nk1, nkm, args.n_ot_slab).total > 160u * 1024u) ft = 1;
        const LdsPlan lp = lcrc_lds_plan(ft, p.nbanks, nk1, nkm, args.n_ot_slab);
        if (lp.total > 160u * 1024u) return hipErrorInvalidValue;
        e = grant_lds(v->h2[ft - 1], vi, 4 + ft, dev);
        if (e != hipSuccess) return e;
        if (variant_name) *variant_name = v->name;
        const int bm = 16 * ft;
        return hipLaunchKernel(v->h2[ft - 1], dim3((rows + bm - 1) / bm), block, kargs, lp.total, stream);
    }

    // ---- launch plan --------------------------------------------------------------------------------------------
    // One launch of 32-frame workgroups costs a whole "round" (every CU one workgroup, ~t32) per n_cu * 32 rows or part
    // thereof; 16-frame workgroups cost half a round each (~t32 / 2) per n_cu * 16 rows; and up to n_cu / 2 tiles the
    // split-hidden path is cheaper still.  So a launch is cut into a MAIN part of whole rounds and a TAIL that takes
    // the cheapest form for its size -- e.g. 12 288 rows = 8192 as 32-frame tiles + 4096 as 16-frame tiles (1.5
    // rounds instead of 2), 4100 rows = 4096 as 16-frame tiles + 4 on the split path.  Both fused forms give the same
    // bits, so cutting never changes a frame's result; the split tail exists only with lcrc_set_hidden_split(h, 0).
    const bool probes = p.dbg_in0 || p.dbg_in1 || p.dbg_p0 || p.dbg_p1 || p.dbg_g;
    const int round32 = n_cu * 32, round16 = n_cu * 16;
    struct Part { int first, count, ft; bool split; };
    Part parts[3];
    int n_parts = 0;
    const bool free_choice = p.tile_frames == 0 && fits32 && !probes && !p.stamps;
    // Whole rounds run as PAIRS of 16-frame workgroups per CU where two of them fit side by side (every shipped shape:
    // <= 256 registers, <= 80 KB of LDS): the same rows per round as one 32-frame workgroup per CU, but while one of the
    // pair is in a phase the f32 MFMA cannot hide -- staging, projection, the softmax / ln() epilogues: 6 % of a CZ
    // workgroup, 15-19 % of an EN one -- the other's hidden loops have the matrix pipe.  Measured (round 4, same bits):
    // CZ 8192 rows 0.1945 -> 0.1925 ms, 32768 0.7738 -> 0.7622; HU 32768 0.9379 -> 0.9253; RU 0.7825 -> 0.7670; EN 8192
    // 0.0803 -> 0.0786, 32768 0.3151 -> 0.3056 (0.731 -> 0.754 of peak).  (Rounds 1 and 2 found the opposite: the weight
    // loads' 64-bit per-lane addresses then cost twice as much per MFMA in 16-frame workgroups; with scalar-base loads they
    // do not.)
    const bool pair16 = 2u * lcrc_lds_plan(1, p.nbanks, k1, km, lcrc_n_ot_slab(p.net)).total <= 160u * 1024u;
    const int ft_round = pair16 ? 1 : 2;
    const int tiles16_all = (rows + 15) / 16;
    const bool small_splits = p.tile_frames != 32 && choose_split(p, tiles16_all, n_cu) > 1;
    if (!free_choice || rows <= round16 || small_splits) {
        // forced tile size, probes, or a launch that fits one half round: a single part (as ever)
        int ft = p.tile_frames == 16 ? 1 : p.tile_frames == 32 ? 2 : ((rows + 31) / 32 <= n_cu / 2 ? 1 : 2);
        if (!fits32 || probes) ft = 1;
        parts[n_parts++] = Part{args.row_first, rows, ft, small_splits};
    } else {
        const int main_rows = rows / round32 * round32;
        const int rem = rows - main_rows;
        // would the split path take a tail of `r` rows?
        auto splits = [&](int r) { return r > 0 && choose_split(p, (r + 15) / 16, n_cu) > 1; };
        if (rem == 0) {
            parts[n_parts++] = Part{args.row_first, rows, ft_round, false};
        } else if (rem <= round16) {
            // whole rounds, then half a round (or less) of 16-frame tiles / the split path; rounds of 16-frame pairs and a
            // fused tail of 16-frame tiles are one launch
            if (splits(rem) || ft_round != 1) {
                if (main_rows > 0) parts[n_parts++] = Part{args.row_first, main_rows, ft_round, false};
                parts[n_parts++] = Part{args.row_first + main_rows, rem, 1, splits(rem)};
            } else {
                parts[n_parts++] = Part{args.row_first, rows, 1, false};
            }
        } else if (splits(rem - round16)) {
            // between half a round and a round, and what lies beyond the half round is small enough for the split
            // path: half a round of 16-frame tiles + the split tail beat the full round
            if (ft_round == 1) {
                parts[n_parts++] = Part{args.row_first, main_rows + round16, 1, false};      // rounds + the half round: one launch
            } else {
                if (main_rows > 0) parts[n_parts++] = Part{args.row_first, main_rows, 2, false};
                parts[n_parts++] = Part{args.row_first + main_rows, round16, 1, false};
            }
            parts[n_parts++] = Part{args.row_first + main_rows + round16, rem - round16, 1, true};
        } else {
            parts[n_parts++] = Part{args.row_first, rows, ft_round, false};
        }
    }
    float *const post0 = args.post;
    const int row_first0 = args.row_first;
    const size_t n_out = (size_t)p.net[2].n_out;
    for (int k = 0; k < n_parts; k++) {
        const Part &pt = parts[k];
        args.row_first = pt.first;
        args.row_end = pt.first + pt.count;
        args.post = post0 + (size_t)(pt.first - row_first0) * n_out;
        const int tiles16 = (pt.count + 15) / 16;
        const int split = pt.split ? choose_split(p, tiles16, n_cu) : 1;
        if (split > 1) {
            // ---- split-hidden path: two launches (band phase, merger phase) on 16-frame tiles ----
            const int nht_b = max(p.net[0].nht, p.net[1].nht), nht_m = p.net[2].nht;
            // band phase: `split` workgroups per tile = split / 2 slices per band net (each workgroup runs ONE net on its four
            // waves); merger phase: `split` slices.  At least one hidden tile per wave.
            const int sb = max(1, min(split / 2, nht_b / 4)), sm = max(1, min(split, nht_m / 4));
            args.tps_b = (nht_b + sb - 1) / sb;
            args.split_b = (nht_b + args.tps_b - 1) / args.tps_b;      // no empty slices
            args.tps_m = (nht_m + sm - 1) / sm;
            args.split_m = (nht_m + args.tps_m - 1) / args.tps_m;
            const LdsPlan lp = lcrc_lds_plan(1, p.nbanks, k1, km, args.n_ot_slab);
            e = grant_lds(v->split_band, vi, 2, dev);
            if (e != hipSuccess) return e;
            e = grant_lds(v->split_merger, vi, 3, dev);
            if (e != hipSuccess) return e;
            e = hipLaunchKernel(v->split_band, dim3(tiles16 * 2 * args.split_b), block, kargs, lp.total + 16, stream);
            if (e != hipSuccess) return e;
            e = hipLaunchKernel(v->split_merger, dim3(tiles16 * args.split_m), block, kargs,
                                lcrc_split_merger_lds(km, args.n_ot_slab), stream);
            if (e != hipSuccess) return e;
            continue;
        }
        // 32-frame workgroups load every weight fragment once per 32 frames; 16-frame workgroups twice as often, but
        // there are twice as many of them: they win while the 32-frame grid would leave at least half of the CUs idle
        const int ft = pt.ft;
        const void *fn = probes ? v->probe : v->fn[ft - 1];
        e = grant_lds(fn, vi, probes ? 4 : ft - 1, dev);
        if (e != hipSuccess) return e;
        const LdsPlan lp = lcrc_lds_plan(ft, p.nbanks, k1, km, lcrc_n_ot_slab(p.net));
        const int bm = 16 * ft;
        e = hipLaunchKernel(fn, dim3((pt.count + bm - 1) / bm), block, kargs, lp.total, stream);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

// Brings this file's code object onto the current device (the runtime loads a file's kernels the first time one of them is
// asked about or launched: 10-25 ms for this one, which otherwise falls into a process's first launch).
hipError_t lcrc_preload_code()
{
    hipFuncAttributes a;
    return hipFuncGetAttributes(&a, kVariants[0].fn[0]);
}

}  // namespace phnrec
