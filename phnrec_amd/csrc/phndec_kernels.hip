// phndec_kernels.hip -- the phoneme-loop Viterbi decoder of the shipped configs (decoder/type=phndec) on
// the GPU: SURVEY.md 8 "next" row f3.  OPTIONAL: BASELINE.json's north star keeps decoding on the host, and
// the CLI does so unless -D is given; this kernel exists so that a `-t str` run needs neither the
// 4*nOut bytes per frame of posteriors over PCIe nor host cores for decoding.
//
// One wave per utterance (four utterances per workgroup), lane i = phoneme i (<= 64 phonemes), each lane carrying
// its model's S+1 token slots (score, entry winner, length) in registers; the time_pruning+1 <= 64 entries of the
// winner history live one per lane in five more registers (v_readlane with a scalar index, compare + select to
// write), wave maxima go through DPP: the kernel touches no LDS and needs no barrier.  Restates, operation by
// operation in f32:
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

// The maxima and the history below are written as gfx9 wave64 instructions (DPP row_bcast does not exist from gfx10 on; a
// 32-wide wave would need other masks): refuse to build for anything else rather than compute something else.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__) && !defined(__gfx942__) && !defined(__gfx90a__)
#error "phndec_kernels.hip is written for wave64 gfx9-class ISA (gfx950; gfx942 / gfx90a share the instructions it uses)"
#endif
#if defined(__HIP_DEVICE_COMPILE__) && defined(__AMDGCN_WAVEFRONT_SIZE__)
static_assert(__AMDGCN_WAVEFRONT_SIZE__ == 64, "one utterance per 64-lane wave");
#endif

namespace phnrec {

namespace {

constexpr int kMaxStates = 4;
constexpr int kMaxHist = 64;       // time_pruning + 1 <= 64: history slot q lives in lane q of three registers

// Maximum over the wave and the LOWEST lane holding it == the reference's first strict maximum.
// Six v_max_f32 with a DPP source operand -- quad_perm, quad_perm, row_half_mirror, row_mirror give every lane its row's
// maximum, row_bcast:15 (rows 1 and 3) and row_bcast:31 (rows 2 and 3) carry it across the rows into lane 63 -- and one
// v_readlane; the lane by ballot + find-first.  No LDS traffic (a __shfl butterfly is six dependent ds_bpermute round trips).
// Written as instructions: through the builtins (update_dpp on the bit pattern + fmaxf) every step was a v_mov_dpp, a
// v_max x, x, x that quiets a possible signalling NaN and the v_max itself, and the four rows met through four v_readlane
// and three more v_max -- 30 instructions per maximum, two maxima per frame on the decoder's one dependent chain.  (The
// values are finite floats or -FLT_MAX: no NaN to quiet.  s_nop 1: a DPP operand written by the previous VALU instruction
// needs two wait states, and the assembler does not see inside an asm block; s_nop 4 in front of the FIRST one: the
// instruction ahead of the block is the compiler's and may be a VALU write of EXEC (v_cmpx), which a DPP read must follow
// by five wait states.)
__device__ __forceinline__ void wave_argmax(float v, float &best, int &lane_of_best)
{
    float m = v;
    asm volatile("s_nop 4\n\tv_max_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf"
                 : "+v"(m));
    best = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, m), 63));
    const unsigned long long hit = __ballot(v == best);
    lane_of_best = hit ? __builtin_ctzll(hit) : 0;
}

__device__ __forceinline__ int lane_get(int v, int lane) { return __builtin_amdgcn_readlane(v, lane); }
__device__ __forceinline__ float lane_get(float v, int lane)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}
}  // namespace

// kDecWaves utterances per workgroup, one wave each, no barrier and no LDS: the waves of a workgroup have nothing to do
// with each other.  What the grouping and the register budget buy is CU time for the posterior kernel, whose workgroups
// fill a CU's register file almost entirely (one 32-frame workgroup: 4 waves x <= 254 registers; since round 4 a PAIR of
// 16-frame workgroups: 8 waves, 384-448 of the 512 registers per SIMD lane).  Round 3's one-wave workgroups landed on 36
// different CUs and kept posterior workgroups off all of them for the 1.2 ms of a 36-utterance launch.  Now a workgroup is
// FOUR waves -- one per SIMD -- of at most 64 registers: it fits into what a pair of posterior workgroups leaves free
// (448 + 64 = 512), so the decoder runs BESIDE the posterior kernel's workgroups instead of in place of them.  (Sixteen waves
// per workgroup, four per SIMD, need 256 registers per SIMD lane: beside one 32-frame workgroup that fitted, beside a pair
// it does not, and -F -D fell 10 % behind -F until this was changed.)
constexpr int kDecWaves = 4;

template <int S>
__global__ __launch_bounds__(64 * kDecWaves, 8) void phndec_kernel(const PhnDecParams p)      // 8 waves per SIMD: <= 64 registers
{
    // (readfirstlane: the wave number is the same in all 64 lanes, which hipcc cannot know -- without it the utterance's
    //  bounds, the history head and every other wave-uniform value live in vector registers and each scalar-indexed
    //  readlane costs a v_readfirstlane and its hazard no-ops)
    const int u = blockIdx.x * kDecWaves + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    if (u >= p.n_utts) return;                                  // wave-uniform
#ifdef PHNDEC_PRIO
    __builtin_amdgcn_s_setprio(PHNDEC_PRIO);
#endif
    const int a0 = p.off[u], T = p.off[u + 1] - a0;
    const int P = p.P, H = p.prune + 1;
    const float lh = -0.69314718055994530941723212145818f;      // ln 0.5, both transitions (phndec.cpp:9,14-15)
    const bool active = lane < P;
    lcrc_label *out = p.labels + a0;
    // winner history (phndec.cpp's hphn / hlen / halpha): physical slot = lane, logical slot q = (head + q) % H.
    // hmask (round 4): bit d of slot q's 64-bit word says that TimePruning's walk, started at q, visits q - d --
    // the walk "offs -= hlen[offs]" (phndec.cpp:206-214) is a chain of back pointers that never changes once a slot is
    // pushed, so the set of positions it visits is built when the slot is pushed, 1 | mask[q - hlen] << hlen, and the
    // per-frame walk (3-8 dependent readlane round trips) becomes two readlanes and a few scalar bit operations.
    // Distances are relative, so the words stay valid while the logical indices slide.
    int hphn = -1, hlen = -1, hm_lo = 0, hm_hi = 0;
    float halpha = -1.0f;
    // a lane's token slots: pad lanes (>= P) carry -FLT_MAX and harmless bookkeeping; they never win a strict compare
    float a[S + 1];
    int pv[S + 1], ln[S + 1];
#pragma unroll
    for (int j = 0; j <= S; j++) { a[j] = j == 0 && active ? p.wpen : -FLT_MAX; pv[j] = -1; ln[j] = 0; }
    float prev_alpha = 0.0f;
    int nlab = 0, head = 0;
    auto phys = [&](int q) { const int x = head + q; return x >= H ? x - H : x; };   // uniform

    // One frame needs S observations per lane; fetched just in time each would cost an L2 / HBM round trip
    // per frame (the frames are strictly sequential), so they are requested kAhead frames ahead into a
    // register ring whose slots are compile-time (the loop is unrolled by kAhead).
    constexpr int kAhead = 8;
    float ob[kAhead][S];
    const int lidx = min(lane, P - 1) * S;
    auto fetch = [&](int t, float (&o)[S]) {                 // unconditional loads, clamped indices
        const float *f = p.logpost + (size_t)(a0 + max(0, min(t, T - 1))) * p.cols + lidx;
#pragma unroll
        for (int j = 0; j < S; j++) o[j] = f[j];
    };
    auto step = [&](int t, const float (&o)[S]) {
        // inside the models, last state first, on the old values (phndec.cpp:96-119); pad lanes compute along
#pragma unroll
        for (int j = S; j > 0; j--) {
            const float stay = a[j] + lh, enter = a[j - 1] + lh;
            const bool keep = stay > enter;
            a[j] = (keep ? stay : enter) + o[j - 1];
            pv[j] = keep ? pv[j] : pv[j - 1];
            ln[j] = (keep ? ln[j] : ln[j - 1]) + 1;
        }
        if (!active) {                                          // -FLT_MAX + x is -FLT_MAX for every finite x here; keep it exact
#pragma unroll
            for (int j = 1; j <= S; j++) a[j] = -FLT_MAX;
        }
        // the best inner token of each lane (TimePruning, phndec.cpp:191-205: first strict maximum = smallest state)
        float bv = a[1];
        int bl = ln[1], bp = pv[1];
#pragma unroll
        for (int j = 2; j <= S; j++)
            if (a[j] > bv) { bv = a[j]; bl = ln[j]; bp = pv[j]; }
        // exit tokens: slot S of every phoneme
        float best, wv;
        int bi, wi;
        wave_argmax(a[S], best, bi);
        wave_argmax(bv, wv, wi);
        const int epv = lane_get(pv[S], bi), eln = lane_get(ln[S], bi);
        head = head + 1 == H ? 0 : head + 1;                    // shift the history left by one ...
        {                                                       // ... and push the winner at the back
            const int back = phys(H - 1);
            // positions the walk from the new slot visits: itself, then whatever the slot eln frames back visits
            unsigned long long m = 1ull;
            if (eln >= 1 && eln <= H - 1) {
                const int src = phys(H - 1 - eln);
                const unsigned long long sm = ((unsigned long long)(unsigned)lane_get(hm_hi, src) << 32) | (unsigned)lane_get(hm_lo, src);
                m |= sm << eln;                                 // eln <= 63
            }
            // the slot's five registers take their wave-uniform values in lane `back`: one s_mov to M0 (the lane select;
            // value and lane in two scalar registers would be two constant-bus reads) and five v_writelane_b32 -- as compare +
            // move out of the scalar register + select it was eleven instructions.  (M0 is not allocatable and nothing else in
            // this kernel uses it -- no LDS, no GDS, no s_movrel --, so writing it here disturbs nothing.)
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
            asm volatile("s_mov_b32 m0, %10\n\t"
                         "v_writelane_b32 %0, %5, m0\n\t"
                         "v_writelane_b32 %1, %6, m0\n\t"
                         "v_writelane_b32 %2, %7, m0\n\t"
                         "v_writelane_b32 %3, %8, m0\n\t"
                         "v_writelane_b32 %4, %9, m0"
                         : "+v"(hphn), "+v"(hlen), "+v"(halpha), "+v"(hm_lo), "+v"(hm_hi)
                         : "s"(epv), "s"(eln), "s"(best), "s"((int)(unsigned)m), "s"((int)(unsigned)(m >> 32)), "s"(back)
                         : "m0");
#pragma clang diagnostic pop
        }
        a[0] = active ? best + p.wpen : -FLT_MAX;
        pv[0] = bi;
        ln[0] = 0;

        const int nframes = t + 1;
        if (nframes >= H) {                                     // TimePruning
            int blen = lane_get(bl, wi), bprev = lane_get(bp, wi);
            if (!(wv > -FLT_MAX)) { blen = 1; bprev = 0; }      // no token beat the initial -FLT_MAX
            const int s0 = H - 1 - blen;                        // where the walk starts; < 0: it ends below the horizon
            bool hit = s0 == 0;
            int phn = bprev;
            if (s0 > 0) {
                const int q = phys(s0);
                const unsigned long long m = ((unsigned long long)(unsigned)lane_get(hm_hi, q) << 32) | (unsigned)lane_get(hm_lo, q);
                hit = (m >> s0) & 1ull;                         // the walk lands exactly on the horizon
                if (hit) {
                    // the last slot it visits above the horizon names the phoneme that ends there
                    const unsigned long long above = m & ((1ull << s0) - 1ull);      // never 0: bit 0 is the start
                    const int d = 63 - __builtin_clzll(above);
                    phn = lane_get(hphn, phys(s0 - d));
                }
            }
            if (hit) {                                          // a phoneme ends exactly at the horizon
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

    // Done(): the winner that entered the loop last, traced back through the history.  The walk produces the tail newest
    // first; `labels` may be pinned HOST memory (the launch code lets the kernel store its results where the host reads them:
    // no copy commands), which is written once and never read here -- so the walk runs twice, first to count the tail's
    // labels, then to store each one at its final place.
    const int tail_phn = lane_get(pv[0], 0);
    int ntail = 0;
    {
        int offs = H - 1, phn = tail_phn;
        while (offs > 0 && phn != -1) {
            const int q = phys(offs);
            const int len = lane_get(hlen, q);
            if (len <= 0) break;
            phn = lane_get(hphn, q);
            offs -= len;
            ntail++;
        }
    }
    {
        int offs = H - 1, end = T, phn = tail_phn, k = 0;
        while (offs > 0 && phn != -1) {
            const int q = phys(offs);
            const int len = lane_get(hlen, q), start = end - len;
            const float al = lane_get(halpha, q);
            const int pphn = lane_get(hphn, q);
            if (len <= 0) break;
            offs -= len;
            const float like = offs > 0 ? al - lane_get(halpha, phys(offs)) : al - prev_alpha;
            if (lane == 0) {
                lcrc_label &l = out[nlab + ntail - 1 - k];
                l.start = start; l.end = end; l.phn = phn; l.score = like;
            }
            k++;
            end = start;
            phn = pphn;
        }
    }
    if (lane == 0) p.count[u] = nlab + ntail;
}

// this file's code object onto the current device ahead of the first decoder launch (lcrc_device_warmup)
hipError_t phndec_preload_code()
{
    hipFuncAttributes a;
    return hipFuncGetAttributes(&a, reinterpret_cast<const void *>(&phndec_kernel<3>));
}

hipError_t phndec_launch(const PhnDecParams &p, hipStream_t stream)
{
    if (p.n_utts <= 0) return hipSuccess;
    if (p.P < 1 || p.P > 64 || p.S < 1 || p.S > kMaxStates || p.prune < 1 || p.prune + 1 > kMaxHist ||
        p.P * p.S > p.cols)
        return hipErrorInvalidValue;
    const dim3 grid((p.n_utts + kDecWaves - 1) / kDecWaves), block(64 * kDecWaves);
    switch (p.S) {
    case 1: phndec_kernel<1><<<grid, block, 0, stream>>>(p); break;
    case 2: phndec_kernel<2><<<grid, block, 0, stream>>>(p); break;
    case 3: phndec_kernel<3><<<grid, block, 0, stream>>>(p); break;
    default: phndec_kernel<4><<<grid, block, 0, stream>>>(p); break;
    }
    return hipGetLastError();
}

}  // namespace phnrec
