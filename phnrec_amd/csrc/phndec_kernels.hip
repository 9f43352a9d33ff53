// phndec_kernels.hip -- the phoneme-loop Viterbi decoder of the shipped configs (decoder/type=phndec) on
// the GPU: SURVEY.md 8 "next" row f3.  OPTIONAL: BASELINE.json's north star keeps decoding on the host, and
// the CLI does so unless -D is given; this kernel exists so that a `-t str` run needs neither the
// 4*nOut bytes per frame of posteriors over PCIe nor host cores for decoding.
//
// One wave per utterance, lane i = phoneme i (<= 64 phonemes), each lane carrying its model's S+1 token
// slots (score, entry winner, length) in registers; the time_pruning+1 <= 64 entries of the winner history
// live one per lane in three more registers (v_readlane / v_writelane with a scalar index), wave maxima go
// through DPP: the kernel touches no LDS and needs no barrier.  Restates, operation by operation in f32:
//   PhnDec::Init          phndec.cpp:44-94     entry slot = insertion penalty, the rest -FLT_MAX
//   PhnDec::ProcessFrame  phndec.cpp:96-189    inside the models last state first (stay vs enter, ln 0.5
//                                              each, strict >), best exit token = first strict maximum,
//                                              history push, re-entry of every phoneme
//   TimePruning           phndec.cpp:191-234   best token of all states -> walk the winner history back to
//                                              the pruning horizon; a boundary exactly there emits a label
//   PhnDec::Done          phndec.cpp:236-303   trace the rest back
// Frames are strictly sequential; the parallelism is across phonemes (lanes) and utterances (waves).
#include <hip/hip_runtime.h>

#include <cfloat>

#include "lcrc_dev.h"

namespace phnrec {

namespace {

constexpr int kMaxStates = 4;
constexpr int kMaxHist = 64;       // time_pruning + 1 <= 64: history slot q lives in lane q of three registers

// Maximum over the wave and the LOWEST lane holding it == the reference's first strict maximum.
// Row maxima by DPP (quad_perm, quad_perm, row_half_mirror, row_mirror), the four rows by v_readlane, the
// lane by ballot + find-first: no LDS traffic (a __shfl butterfly is six dependent ds_bpermute round trips).
__device__ __forceinline__ void wave_argmax(float v, float &best, int &lane_of_best)
{
    float m = v;
    m = fmaxf(m, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, m), 0xB1, 0xF, 0xF, true)));
    m = fmaxf(m, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, m), 0x4E, 0xF, 0xF, true)));
    m = fmaxf(m, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, m), 0x141, 0xF, 0xF, true)));
    m = fmaxf(m, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, m), 0x140, 0xF, 0xF, true)));
    const int mi = __builtin_bit_cast(int, m);
    const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(mi, 0));
    const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(mi, 16));
    const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(mi, 32));
    const float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(mi, 48));
    best = fmaxf(fmaxf(r0, r1), fmaxf(r2, r3));
    const unsigned long long hit = __ballot(v == best);
    lane_of_best = hit ? __builtin_ctzll(hit) : 0;
}

__device__ __forceinline__ int lane_get(int v, int lane) { return __builtin_amdgcn_readlane(v, lane); }
__device__ __forceinline__ float lane_get(float v, int lane)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}
// lane `lane` of the register := v (both wave-uniform): a compare + select
__device__ __forceinline__ int lane_set(int old, int lane, int v) { return (int)threadIdx.x == lane ? v : old; }
__device__ __forceinline__ float lane_set(float old, int lane, float v) { return (int)threadIdx.x == lane ? v : old; }

}  // namespace

__global__ __launch_bounds__(64) void phndec_kernel(const PhnDecParams p)
{
    const int u = blockIdx.x, lane = threadIdx.x;
    const int a0 = p.off[u], T = p.off[u + 1] - a0;
    const int S = p.S, P = p.P, H = p.prune + 1;
    const float lh = -0.69314718055994530941723212145818f;      // ln 0.5, both transitions (phndec.cpp:9,14-15)
    const bool active = lane < P;
    lcrc_label *out = p.labels + a0;

    // winner history (phndec.cpp's hphn / hlen / halpha): physical slot = lane, logical slot q = (head + q) % H
    int hphn = -1, hlen = -1;
    float halpha = -1.0f;
    float a[kMaxStates + 1];
    int pv[kMaxStates + 1], ln[kMaxStates + 1];
#pragma unroll
    for (int j = 0; j <= kMaxStates; j++) { a[j] = j == 0 ? p.wpen : -FLT_MAX; pv[j] = -1; ln[j] = 0; }
    float prev_alpha = 0.0f;
    int nlab = 0, head = 0;
    auto phys = [&](int q) { const int x = head + q; return x >= H ? x - H : x; };   // uniform

    // One frame needs S observations per lane; fetched just in time each would cost an L2 / HBM round trip
    // per frame (the frames are strictly sequential), so they are requested kAhead frames ahead into a
    // register ring whose slots are compile-time (the loop is unrolled by kAhead).
    constexpr int kAhead = 8;
    float ob[kAhead][kMaxStates];
    const int lidx = min(lane, P - 1) * S;
    auto fetch = [&](int t, float (&o)[kMaxStates]) {        // unconditional loads, clamped indices
        const float *f = p.logpost + (size_t)(a0 + max(0, min(t, T - 1))) * p.cols + lidx;
#pragma unroll
        for (int j = 0; j < kMaxStates; j++) o[j] = f[min(j, S - 1)];
    };
    auto step = [&](int t, const float (&o)[kMaxStates]) {
        if (active) {
#pragma unroll
            for (int j = kMaxStates; j > 0; j--) {
                if (j <= S) {
                    const float stay = a[j] + lh, enter = a[j - 1] + lh;
                    const float obs = o[j - 1];
                    if (stay > enter) {
                        a[j] = stay + obs;
                        ln[j] += 1;
                    } else {
                        a[j] = enter + obs;
                        pv[j] = pv[j - 1];
                        ln[j] = ln[j - 1] + 1;
                    }
                }
            }
        }
        // exit tokens: slot S of every phoneme
        float ex = -FLT_MAX;
        int epv = -1, eln = 0;
#pragma unroll
        for (int j = 1; j <= kMaxStates; j++)
            if (j == S) { ex = active ? a[j] : -FLT_MAX; epv = pv[j]; eln = ln[j]; }
        float best;
        int bi;
        wave_argmax(ex, best, bi);
        epv = lane_get(epv, bi);
        eln = lane_get(eln, bi);
        head = head + 1 == H ? 0 : head + 1;                    // shift the history left by one ...
        {                                                       // ... and push the winner at the back
            const int back = phys(H - 1);
            hphn = lane_set(hphn, back, epv);
            hlen = lane_set(hlen, back, eln);
            halpha = lane_set(halpha, back, best);
        }
        a[0] = best + p.wpen;
        pv[0] = bi;
        ln[0] = 0;

        const int nframes = t + 1;
        if (nframes >= H) {                                     // TimePruning
            float bv = -FLT_MAX;
            int bl = 1, bp = 0;
#pragma unroll
            for (int j = 1; j <= kMaxStates; j++)
                if (j <= S && active && a[j] > bv) { bv = a[j]; bl = ln[j]; bp = pv[j]; }
            float wv;
            int wi;
            wave_argmax(bv, wv, wi);
            int blen = lane_get(bl, wi), bprev = lane_get(bp, wi);
            if (!(wv > -FLT_MAX)) { blen = 1; bprev = 0; }      // no token beat the initial -FLT_MAX
            int offs = H - 1 - blen, phn = bprev;
            while (offs > 0) {
                const int q = phys(offs);
                const int l = lane_get(hlen, q);
                phn = lane_get(hphn, q);
                if (l <= 0) break;
                offs -= l;
            }
            if (offs == 0) {                                    // a phoneme ends exactly at the horizon
                const int q0 = phys(0);
                const int end = nframes - H + 1, start = end - lane_get(hlen, q0);
                const float h0 = lane_get(halpha, q0);
                const float like = h0 - prev_alpha;
                prev_alpha = h0;
                if (phn >= 0) {
                    if (lane == 0) { out[nlab].start = start; out[nlab].end = end; out[nlab].phn = phn; out[nlab].score = like; }
                    nlab++;
                }
            }
        }
    };
    if (T > 0) {
#pragma unroll
        for (int d = 0; d < kAhead; d++) fetch(d, ob[d]);
        for (int t0 = 0; t0 < T; t0 += kAhead) {
#pragma unroll
            for (int d = 0; d < kAhead; d++) {
                if (t0 + d < T) step(t0 + d, ob[d]);
                fetch(t0 + d + kAhead, ob[d]);
            }
        }
    }

    // Done(): the winner that entered the loop last, traced back through the history
    int offs = H - 1, end = T, phn = lane_get(pv[0], 0), ntail = 0;
    while (offs > 0 && phn != -1) {
        const int q = phys(offs);
        const int len = lane_get(hlen, q), start = end - len;
        const float al = lane_get(halpha, q);
        const int pphn = lane_get(hphn, q);
        if (len <= 0) break;
        offs -= len;
        const float like = offs > 0 ? al - lane_get(halpha, phys(offs)) : al - prev_alpha;
        if (lane == 0) {
            lcrc_label &l = out[nlab + ntail];
            l.start = start; l.end = end; l.phn = phn; l.score = like;
        }
        ntail++;
        end = start;
        phn = pphn;
    }
    if (lane == 0) {
        for (int i = 0, j = ntail - 1; i < j; i++, j--) {       // the tail was produced newest first
            const lcrc_label tmp = out[nlab + i];
            out[nlab + i] = out[nlab + j];
            out[nlab + j] = tmp;
        }
        p.count[u] = nlab + ntail;
    }
}

hipError_t phndec_launch(const PhnDecParams &p, hipStream_t stream)
{
    if (p.n_utts <= 0) return hipSuccess;
    if (p.P < 1 || p.P > 64 || p.S < 1 || p.S > kMaxStates || p.prune < 1 || p.prune + 1 > kMaxHist ||
        p.P * p.S > p.cols)
        return hipErrorInvalidValue;
    phndec_kernel<<<dim3(p.n_utts), dim3(64), 0, stream>>>(p);
    return hipGetLastError();
}

}  // namespace phnrec
