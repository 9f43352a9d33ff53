// phndec_kernels.hip -- the phoneme-loop Viterbi decoder of the shipped configs (decoder/type=phndec) on
// the GPU: SURVEY.md 8 "next" row f3.  OPTIONAL: BASELINE.json's north star keeps decoding on the host, and
// the CLI does so unless -D is given; this kernel exists so that a `-t str` run needs neither the
// 4*nOut bytes per frame of posteriors over PCIe nor host cores for decoding.
//
// One wave per utterance, lane i = phoneme i (<= 64 phonemes), each lane carrying its model's S+1 token
// slots (score, entry winner, length) in registers.  Restates, operation by operation in f32:
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
constexpr int kMaxHist = 256;      // time_pruning + 1 <= 256

// (value, index) maximum with the LOWEST index among equal values == the reference's first strict maximum
__device__ __forceinline__ void wave_argmax(float &v, int &idx)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const float ov = __shfl_xor(v, d);
        const int oi = __shfl_xor(idx, d);
        if (ov > v || (ov == v && oi < idx)) { v = ov; idx = oi; }
    }
}

}  // namespace

__global__ __launch_bounds__(64) void phndec_kernel(const PhnDecParams p)
{
    __shared__ int hphn[kMaxHist], hlen[kMaxHist];
    __shared__ float halpha[kMaxHist];
    const int u = blockIdx.x, lane = threadIdx.x;
    const int a0 = p.off[u], T = p.off[u + 1] - a0;
    const int S = p.S, P = p.P, H = p.prune + 1;
    const float lh = -0.69314718055994530941723212145818f;      // ln 0.5, both transitions (phndec.cpp:9,14-15)
    const bool active = lane < P;
    lcrc_label *out = p.labels + a0;

    for (int q = lane; q < H; q += 64) { hphn[q] = -1; hlen[q] = -1; halpha[q] = -1.0f; }
    float a[kMaxStates + 1];
    int pv[kMaxStates + 1], ln[kMaxStates + 1];
#pragma unroll
    for (int j = 0; j <= kMaxStates; j++) { a[j] = j == 0 ? p.wpen : -FLT_MAX; pv[j] = -1; ln[j] = 0; }
    float prev_alpha = 0.0f;
    int nlab = 0, head = 0;         // logical history slot q lives at (head + q) % H
    __syncthreads();

    for (int t = 0; t < T; t++) {
        const float *f = p.logpost + (size_t)(a0 + t) * p.cols;
        if (active) {
#pragma unroll
            for (int j = kMaxStates; j > 0; j--) {
                if (j <= S) {
                    const float stay = a[j] + lh, enter = a[j - 1] + lh;
                    const float obs = f[lane * S + (j - 1)];
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
        float best = ex;
        int bi = lane;
        wave_argmax(best, bi);
        epv = __shfl(epv, bi);
        eln = __shfl(eln, bi);
        head = head + 1 == H ? 0 : head + 1;                    // shift the history left by one ...
        if (lane == 0) {                                        // ... and push the winner at the back
            const int back = head + H - 1 >= H ? head - 1 : head + H - 1;
            hphn[back] = epv; hlen[back] = eln; halpha[back] = best;
        }
        a[0] = best + p.wpen;
        pv[0] = bi;
        ln[0] = 0;
        __syncthreads();

        const int nframes = t + 1;
        if (nframes >= H) {                                     // TimePruning
            float bv = -FLT_MAX;
            int bl = 1, bp = 0;
#pragma unroll
            for (int j = 1; j <= kMaxStates; j++)
                if (j <= S && active && a[j] > bv) { bv = a[j]; bl = ln[j]; bp = pv[j]; }
            float wv = bv;
            int wi = lane;
            wave_argmax(wv, wi);
            int blen = __shfl(bl, wi), bprev = __shfl(bp, wi);
            if (!(wv > -FLT_MAX)) { blen = 1; bprev = 0; }      // no token beat the initial -FLT_MAX
            int offs = H - 1 - blen, phn = bprev;
            while (offs > 0) {
                const int q = head + offs >= H ? head + offs - H : head + offs;
                const int l = hlen[q];
                phn = hphn[q];
                if (l <= 0) break;
                offs -= l;
            }
            if (offs == 0) {                                    // a phoneme ends exactly at the horizon
                const int end = nframes - H + 1, start = end - hlen[head];
                const float like = halpha[head] - prev_alpha;
                prev_alpha = halpha[head];
                if (phn >= 0) {
                    if (lane == 0) { out[nlab].start = start; out[nlab].end = end; out[nlab].phn = phn; out[nlab].score = like; }
                    nlab++;
                }
            }
        }
        __syncthreads();
    }

    // Done(): the winner that entered the loop last, traced back through the history
    int offs = H - 1, end = T, phn = pv[0], ntail = 0;
    while (offs > 0 && phn != -1) {
        const int q = head + offs >= H ? head + offs - H : head + offs;
        const int len = hlen[q], start = end - len;
        const float al = halpha[q];
        const int pphn = hphn[q];
        if (len <= 0) break;
        offs -= len;
        float like;
        if (offs > 0) {
            const int q2 = head + offs >= H ? head + offs - H : head + offs;
            like = al - halpha[q2];
        } else {
            like = al - prev_alpha;
        }
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
