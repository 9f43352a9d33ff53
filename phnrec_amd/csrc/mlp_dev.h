// mlp_dev.h -- device code shared by the fused LCRC kernel (lcrc_kernels.hip) and the kernels of the
// other `posteriors/system` variants (traps_kernels.hip): MFMA wrapper, FEXP, the sigmoid block, the
// ring-form hidden loop, run_net (one MLP on a workgroup's frames incl. fold, softmax and epilogue),
// softening functions.  See lcrc_kernels.hip for the geometry and the arithmetic contract.
#ifndef PHNREC_MLP_DEV_H
#define PHNREC_MLP_DEV_H

#include <hip/hip_runtime.h>

#include <cfloat>
#include <climits>

#include "lcrc_dev.h"

namespace phnrec {

typedef float f4 __attribute__((ext_vector_type(4)));
// the same in the GLOBAL address space, for pointers rebuilt from scalar registers (scalar_ptr)
typedef __attribute__((address_space(1))) const f4 gf4;

template <int V>
struct IntTag { static constexpr int value = V; };

// In-kernel phase stamps exist only in the DIAGNOSTIC build (make stamps ->
// libphnrec_lcrc_stamps.so, tools/stamp_profile.py); in the product the macro is empty.
#ifdef LCRC_STAMPS
#define LCRC_STAMP(p, wave, lane, idx)                                                        \
    do {                                                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                    \
        unsigned long long t_;                                                                \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");           \
        if ((p).stamps && (lane) == 0) (p).stamps[((size_t)blockIdx.x * 8 + (wave)) * 16 + (idx)] = t_; \
        __builtin_amdgcn_sched_barrier(0);                                                    \
    } while (0)
#else
#define LCRC_STAMP(p, wave, lane, idx) do { } while (0)
#endif

__device__ __forceinline__ f4 mfma16x16x4(float a, float b, f4 c)
{
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// ---- FEXP (fexp.h:14-21) ----------------------------------------------------------
// hi32 = (int)(2^20/ln2 * y) + (1072693248 - 60801); lo32 = 0; reinterpret as double.
// (int) is x86 cvttsd2si: INT_MIN when the product is >= 2^31 or NaN.  v_cvt_i32_f64
// saturates instead, so that one case is patched; the test is made on y itself
// (0x1.62e43p+10f = 1419.5654296875 is the smallest float whose product reaches 2^31;
// !(y < T) is also true for NaN).  Below -2^31 both conversions give INT_MIN.
__device__ __forceinline__ double fexp_d(float y)
{
    const double a = 1048576.0 / 0.69314718055994530942;
    int i = __double2int_rz(a * (double)y);
    if (!(y < 0x1.62e43p+10f)) i = INT_MIN;
    unsigned hi = (unsigned)i + 1072632447u;
    return __hiloint2double((int)hi, 0);
}

__device__ __forceinline__ float fexp_f(float y) { return (float)fexp_d(y); }

// FEXP for y <= 0 (the softmax's arguments v_j - max): the product never reaches +2^31, and below -2^31
// v_cvt_i32_f64 saturates to INT_MIN, which is x86's value too -- no patch needed.  (NaN gives 0 where x86
// gives INT_MIN; a NaN row is garbage either way.)
__device__ __forceinline__ float fexp_nonpos_f(float y)
{
    const double a = 1048576.0 / 0.69314718055994530942;
    const unsigned hi = (unsigned)__double2int_rz(a * (double)y) + 1072632447u;
    return (float)__hiloint2double((int)hi, 0);
}

// Loads the first N (1..4) components of a weight fragment.  The last float4 group of
// layer 1 is only partly used when ksteps % 4 != 0; loading all four components would
// leave the unused ones as dead registers with a load in flight, and the first reuse
// of such a register costs an s_waitcnt vmcnt(0) that drains the whole prefetch.
template <int N>
__device__ __forceinline__ f4 load_frag(gf4 *p)
{
    // the components beyond N are never read (the MFMA loop stops at the net's k-steps): they stay UNDEFINED
    // rather than zero -- zeroing them cost a v_mov per component and hidden tile
    if constexpr (N >= 4) {
        return *p;
    } else if constexpr (N == 3) {
        typedef float f3 __attribute__((ext_vector_type(3)));
        typedef __attribute__((address_space(1))) const f3 gf3;
        const f3 t = *(gf3 *)p;
        return __builtin_shufflevector(t, t, 0, 1, 2, -1);
    } else if constexpr (N == 2) {
        typedef float f2 __attribute__((ext_vector_type(2)));
        typedef __attribute__((address_space(1))) const f2 gf2;
        const f2 t = *(gf2 *)p;
        return __builtin_shufflevector(t, t, 0, 1, -1, -1);
    } else {
        typedef float f1 __attribute__((ext_vector_type(1)));
        typedef __attribute__((address_space(1))) const f1 gf1;
        const f1 t = *(gf1 *)p;
        return __builtin_shufflevector(t, t, 0, -1, -1, -1);
    }
}

// A wave-uniform pointer forced into SGPRs: address arithmetic on it runs on the scalar unit, and a load
// `p[imm + lane]` becomes `global_load v, v_lane, s[p] offset:imm` (13-bit immediates: up to 4 fragments of
// 1 KiB per scalar base; beyond that hipcc otherwise forms 64-bit per-lane addresses on the VALU, which the f32
// MFMA shares).
__device__ __forceinline__ gf4 *scalar_ptr(const f4 *p)
{
    // (the integer round trip would lose the address space -- flat loads count on both wait counters and drain
    //  the prefetch -- so the result is typed as a GLOBAL pointer explicitly)
    const unsigned long long u = reinterpret_cast<unsigned long long>(p);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u), hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
    return (gf4 *)(((unsigned long long)hi << 32) | lo);
}

// ---- one MLP on the workgroup's frames -----------------------------------------------
// XF: LDS image of the normalised input, [f][kq][lane] float4 where element j of
//     lane l holds X[frame 16f + (l&15)][k = 16kq + 4j + (l>>4)].
template <int KS, int NKQ, bool EXACT>
__device__ __forceinline__ f4 load_w1_frag(gf4 *t, int kq_local, int kq, int lane)
{
    // `t` (scalar base of a group of four fragments) stays in SGPRs (saddr addressing); only lane*16 is a vector
    // offset; kq_local = position in the group, kq = the fragment's k-group (the last one may be partial)
    return (EXACT && kq == NKQ - 1) ? load_frag<KS - 4 * (NKQ - 1)>(t + kq_local * 64 + lane) : t[kq_local * 64 + lane];
}

// Sigmoid (nn.cpp:796-820) of the 4*FT pre-activations a lane holds, as a sequence of
// STAGES: stage k applies step k to all values, so the dependency chains advance together
// and consecutive VALU instructions are independent.  The hidden loop runs it as one block
// of VALU work in front of a tile's MFMA groups.
//
// Cost matters here: v_mfma_f32_16x16x4_f32 runs on the SIMD's f32 FMA lanes, so VALU
// work does NOT hide behind it (measured, tools/ubench/valu_overlap.hip: every VALU
// instruction adds its full issue time to the 32 cycles of an MFMA).  Hence:
//   * FEXP's integer part stays exact (f64 product, truncation, x86 overflow value);
//   * FEXP's value is exactly representable in f32 (20 mantissa bits), so one
//     v_cvt_f32_f64 of {0, hi} yields it without error in the whole normal range;
//   * 1/(1+e) is evaluated in f32 by v_rcp_f32 alone (1 ulp).  The reference evaluates it in f64 and
//     rounds once; a Newton step + v_div_fixup_f32 would recover the last ulp for 3 more instructions per
//     value -- 1-2 % of the kernel (profiles/r01_ab_runs.txt) for a change of 1e-7 in posteriors whose
//     distance to the reference is 5e-6 either way (summation order of the products).
// Pad hidden units (>= n_hid) need no zeroing: their layer-2 weights are packed as zeros.
template <int FT>
struct SigTile {
    static constexpr int kN = 4 * FT;
    float x[kN];     // -x, then e, 1+e, and finally the sigmoid
    double t[kN];
    static constexpr int kStages = 4;

    __device__ __forceinline__ void begin(const f4 (&p)[FT])
    {
#pragma unroll
        for (int f = 0; f < FT; f++)
#pragma unroll
            for (int i = 0; i < 4; i++) x[4 * f + i] = -p[f][i];
    }
    // `a`: FEXP's constant 2^20 / ln 2 -- times a power of two where the pre-activations arrive scaled (split-f16
    // arithmetic: exact, so the value is the one the unscaled pre-activation gives)
    static constexpr double kFexpA = 1048576.0 / 0.69314718055994530942;
    __device__ __forceinline__ void stage(int k, const double a = kFexpA)
    {
#pragma unroll
        for (int i = 0; i < kN; i++) {
            switch (k) {
            case 0: t[i] = a * (double)x[i]; break;
            case 1: {
                // FEXP's integer hi word (see fexp_d) without the compare / select of the out-of-range patch:
                // i = -(int)trunc(-t).  Truncation is symmetric, so this is (int)trunc(t) in range; for
                // t >= 2^31 the conversion of -t saturates to INT_MIN and -INT_MIN wraps to INT_MIN = x86's
                // cvttsd2si value.  (For t <= -2^31, i.e. a hidden pre-activation above +1419 where the
                // reference's own sigmoid is garbage -- FEXP wraps to -0.97 and 1/(1+e) = 33 --, this gives
                // INT_MIN + 1: that garbage differs in the last of its 20 mantissa bits.)  The negation rides on
                // the instruction's source modifier and on sub instead of add: two VALU instructions per value
                // fewer in the hidden loops.
                const unsigned hi = 1072632447u - (unsigned)__double2int_rz(-t[i]);
                t[i] = __hiloint2double((int)hi, 0);
                break;
            }
            case 2: x[i] = 1.0f + (float)t[i]; break;
            case 3: x[i] = __builtin_amdgcn_rcpf(x[i]); break;      // 1 ulp; 1/inf = 0 and 1/0 = inf natively
            default: break;
            }
        }
    }
    __device__ __forceinline__ void finish(f4 (&s)[FT]) const
    {
#pragma unroll
        for (int f = 0; f < FT; f++)
#pragma unroll
            for (int i = 0; i < 4; i++) s[f][i] = x[4 * f + i];
    }
};

// layer 2: k-slot g of step r is hidden unit 16ht + 4g + r on both operands
template <int FT>
__device__ __forceinline__ void gemm2_group(f4 (&acc)[FT], const f4 &wot, const f4 (&s)[FT])
{
#pragma unroll
    for (int r = 0; r < 4; r++)
#pragma unroll
        for (int f = 0; f < FT; f++) acc[f] = mfma16x16x4(wot[r], s[f][r], acc[f]);
}

// All-reduce over aligned groups of 8 lanes with DPP (no LDS traffic, unlike __shfl_xor which is a
// ds_bpermute): quad_perm [1,0,3,2], quad_perm [2,3,0,1], then row_half_mirror (lane i <-> 7-i, i.e. the
// other quad of the 8-lane half row, which by then holds that quad's result).
// 4 lanes: the two quad_perm steps; 8: + row_half_mirror; 16: + row_mirror (lane i <-> 15-i).
template <int LANES, typename Op>
__device__ __forceinline__ float allreduce(float v, Op op)
{
    v = op(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true)));
    v = op(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true)));
    if constexpr (LANES >= 8)
        v = op(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true)));
    if constexpr (LANES == 16)
        v = op(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true)));
    return v;
}

// Compiler fences (no instruction).  hipcc otherwise SINKS read-only prefetch loads down
// to their first use (IR level) or hoists the register-only MFMA/VALU work of the next
// phase above them (machine scheduler); either way the prefetch becomes a just-in-time
// load and an L2 round trip per fragment is exposed.
#define LCRC_FENCE()                        \
    do {                                    \
        asm volatile("" ::: "memory");      \
        __builtin_amdgcn_sched_barrier(0);  \
    } while (0)

// Softening functions of srec.cpp:164-176 (lcrc_output_configure); constants are formed on the
// host with the reference's f32 expressions.
__device__ __forceinline__ float soften(int func, const float *c, const float *l, float v)
{
    if (func == 1) return logf(v);
    if (func == 2) {
        if (v < c[0]) return logf(v * c[1]) / l[0];
        return -1.0f * logf((1.0f + (-1.0f * v)) * c[2]) / l[1];
    }
    if (func == 3) return sqrtf(-2.0f * logf(v));
    return v;
}

// Row epilogue of a merger: posteriors -> [softening stage 1 -> stage 2 -> byte order] -> row i of `outbuf`
// ([frames][O] floats in LDS, copied to HBM as whole rows afterwards).  One uniform branch per row decides between
// the plain stores and the writer path (per value it was a scalar compare + branch for each of a lane's values).
struct WriterPathEpilogue {
    float *outbuf;
    int O, f0, f1, be, ovalid;
    float c0[3], c1[3], l0[2], l1[2];
    template <int LPF, int NV>
    __device__ __forceinline__ void operator()(int, int i, int part, IntTag<LPF>, const float (&q)[NV], int) const
    {
        float *row = outbuf + i * O + part;
        if ((f0 | f1 | be) == 0) {
#pragma unroll
            for (int j = 0; j < NV; j++)
                if (LPF * j + LPF <= ovalid || part + LPF * j < O) row[LPF * j] = q[j];
        } else {
#pragma unroll
            for (int j = 0; j < NV; j++) {
                float v = soften(f0, c0, l0, q[j]);
                v = soften(f1, c1, l1, v);
                if (be) v = __builtin_bit_cast(float, __builtin_bswap32(__builtin_bit_cast(unsigned, v)));
                if (LPF * j + LPF <= ovalid || part + LPF * j < O) row[LPF * j] = v;
            }
        }
    }
};

// Scatter one value of a net-input row into the MFMA B image.
__device__ __forceinline__ void xf_store(float *img, int nkq, int frame, int k, float v)
{
    const int f = frame >> 4, kq = k >> 4, j = (k >> 2) & 3, l = (frame & 15) + 16 * (k & 3);
    img[((f * nkq + kq) * 64 + l) * 4 + j] = v;
}

// ---- hidden loop, ring form ----------------------------------------------------------------
// One wave per SIMD (a second one does not pay: DESIGN.md 3, "What bounds it").  The weight fragments stream
// through a RING of R registers-quads instead of one buffer per layer.  Per hidden tile t the
// fragments are consumed in the fixed order
//     entry e <  NOT         W2(t)[e]         layer 2 of tile t      (8 MFMAs)
//     entry e <  NOT + NKQ   W1(t+1)[e-NOT]   layer 1 of tile t+1    (8 MFMAs)
//     entry e <  FP          (padding so that FP % R == 0: ring slots are compile-time)
// and entry e lives in slot e % R.  Right after the MFMAs of entry i are issued, the fragment
// of entry i + R (possibly of the next tile) is requested into the slot just consumed: every
// fragment is requested R groups (>= R * 256 cycles) before its use, and at most R requests
// are in flight, so the in-order vmcnt waits never drain more than the one fragment needed.
constexpr int lcrc_ring_size(int f, int n_ot = 0)
{
    int best = 8, pad = 1 << 30;
    // (12 or more output tiles: at most 10 slots -- with 12 the 32-frame variants spill accumulators to AGPRs)
    for (int r = n_ot >= 12 ? 10 : 12; r >= 7; r--) {             // least padding; ties -> the deeper ring
        const int p = (f + r - 1) / r * r - f;
        if (p < pad) { pad = p; best = r; }
    }
    return best;
}

// BKQ: the first BKQ k-groups of the B (input) image are held in REGISTERS for the whole loop instead of being read
// from LDS in every hidden tile (a ds_read_b128 among MFMAs costs the wave ~8.5 cycles, tools/ubench/mfma_shape).
// hipcc does this by itself for the 1500-unit systems' band nets; BKQ makes it explicit where it does not.
template <int KS, int NOT, int FT, bool EXACT, int BKQ = 0>
struct RingLoop {
    static constexpr int NKQ = (KS + 3) / 4;
    static_assert(BKQ <= NKQ, "at most the whole image");
    static constexpr int F = NOT + NKQ;
    static constexpr int R = lcrc_ring_size(F, NOT);
    static constexpr int FP = (F + R - 1) / R * R;
    enum { PRO = 0, MID = 1, LAST = 2 };

    const f4 *w1, *w2;
    const f4 *z1, *z2;          // the all-zero fragment behind w1p / w2p (run-time shapes)
    int zi1, zi2;               // ... as fragment indices: nht * nkq, nht * n_ot
    const float *b1;
    const f4 *XF;
    int hlast, lane;
    int ks, nkq, n_ot;          // run-time sizes (generic shapes); KS / NKQ / NOT when EXACT
    f4 ring[R];
    f4 bimg[BKQ > 0 ? BKQ : 1][FT];

    // request entry e of tile t into slot (e % R); e is a compile-time value after unrolling.
    // Generic shapes: entries beyond the run-time sizes re-request the last valid fragment (a load
    // under a branch would make the compiler drain all requests at the join); they are never consumed.
    __device__ __forceinline__ void request(int slot, int e, int t) { request(slot, e, t, nkq, n_ot); }
    // fragment index `a` if `valid`, else `b`, WITHOUT control flow: as a conditional expression hipcc sinks the address
    // arithmetic into a branch, and one branch per 8-MFMA group cuts the hidden loop into as many basic blocks -- no
    // load or LDS read is then scheduled ahead across them (the run-time-shape loops ran at 59 % MFMA occupancy so)
    static __device__ __forceinline__ int pick_index(bool valid, int a, int b) { return b + (-(int)valid & (a - b)); }
    __device__ __forceinline__ void request(int slot, int e, int t, int nkq, int n_ot)
    {
        if (e < NOT) {
            if (EXACT) {
                const f4 *tb = w2 + (size_t)max(0, min(t, hlast)) * NOT * 64;
                ring[slot] = scalar_ptr(tb + (e & ~3) * 64)[(e & 3) * 64 + lane];
            } else {                                   // past this net's output tiles: the zero fragment behind w2p
                const int idx = pick_index(e < n_ot, max(0, min(t, hlast)) * n_ot + e, zi2);
                ring[slot] = scalar_ptr(w2 + (size_t)idx * 64)[lane];
            }
        } else if (e - NOT < NKQ) {
            const int kq = e - NOT;
            if (EXACT) {
                const f4 *tb = w1 + (size_t)min(t + 1, hlast) * NKQ * 64;
                ring[slot] = load_w1_frag<KS, NKQ, true>(scalar_ptr(tb + (kq & ~3) * 64), kq & 3, kq, lane);
            } else {                                   // past this net's k-groups: the zero fragment behind w1p
                const int idx = pick_index(kq < nkq, min(t + 1, hlast) * nkq + kq, zi1);
                ring[slot] = scalar_ptr(w1 + (size_t)idx * 64)[lane];
            }
        }
    }

    // one pass over the entries of tile t.  PRO: t = first tile - 1, layer-2 entries are skipped
    // (this computes layer 1 of the first tile); LAST: layer 2 only, nothing new is requested
    // beyond this tile's own W2 fragments.
    template <int MODE>
    __device__ __forceinline__ void pass(f4 (&acc)[NOT][FT], f4 (&pre)[FT], f4 &bias, int t)
    {
        const int g = lane >> 4;
        // run-time shapes: the bounds are re-materialised per pass (an empty asm the optimiser cannot see through), so the
        // ~80 comparisons against them are redone on the scalar unit in every pass instead of being hoisted out of the
        // hidden loop into as many SGPR pairs, which spilled
        int nkq = this->nkq, n_ot = this->n_ot;
        if (!EXACT) asm volatile("" : "+s"(nkq), "+s"(n_ot));
        f4 s[FT], nxt[FT];
        if (MODE != PRO) {
            SigTile<FT> sg;
            sg.begin(pre);
#pragma unroll
            for (int k = 0; k < SigTile<FT>::kStages; k++) sg.stage(k);
            sg.finish(s);
        }
#pragma unroll
        for (int f = 0; f < FT; f++) nxt[f] = bias;
        __builtin_amdgcn_sched_barrier(0);
        if (MODE != LAST)                          // bias of the tile after next: requested first, so
            bias = *reinterpret_cast<const f4 *>(b1 + 16 * min(t + 2, hlast) + 4 * g);   // it is old when needed
        f4 xb[2][FT];
        // The B image's row stride is the CLASS's k-groups (NKQ) for run-time shapes too: every read below is then the
        // image base plus a compile-time offset (groups past the net's own are zeros, and so are their weights)
        constexpr int nkq_x = NKQ;
        if (MODE != LAST && BKQ == 0) {
#pragma unroll
            for (int f = 0; f < FT; f++) xb[0][f] = XF[f * nkq_x * 64 + lane];
        }
#pragma unroll
        for (int i = (MODE == PRO ? NOT : 0); i < (MODE == LAST ? NOT : FP); i++) {
            if (i < NOT) {
                gemm2_group<FT>(acc[i], ring[i % R], s);      // (run-time shapes: zero weights past the net's output tiles)
            } else if (i - NOT < NKQ) {
                const int kq = i - NOT;
                if (kq + 1 < NKQ && kq + 1 >= BKQ) {   // B fragments of the next group, ahead of the MFMAs
                    const int kn = kq + 1;
#pragma unroll
                    for (int f = 0; f < FT; f++) xb[(kq + 1) & 1][f] = XF[(f * nkq_x + kn) * 64 + lane];
                }
#pragma unroll
                for (int j = 0; j < 4; j++)
                    // run-time shapes: every k-step of the size class runs (zero weights past the net's last k-step: packed
                    // zeros inside its last group, the zero fragment for whole groups); a conditionally executed MFMA group
                    // costs hipcc's allocation a shuffle of the accumulators, 2x the loop time
                    if (EXACT ? (4 * kq + j < KS) : true) {
#pragma unroll
                        for (int f = 0; f < FT; f++)
                            nxt[f] = mfma16x16x4(ring[i % R][j], kq < BKQ ? bimg[kq < BKQ ? kq : 0][f][j] : xb[kq & 1][f][j], nxt[f]);
                    }
            }
            const int e = (i + R) % FP, dt = (i + R) / FP;
            if (MODE != LAST || (dt == 0 && e < NOT)) request(i % R, e, t + dt, nkq, n_ot);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (MODE != LAST) {
#pragma unroll
            for (int f = 0; f < FT; f++) pre[f] = nxt[f];
        }
        LCRC_FENCE();
    }

    // begin(): the first bias quad and the ring fill for the prologue pass (entries NOT .. NOT+R-1 of pseudo tile
    // ht0 - 1) are REQUESTED; finish() runs the passes.  run() = both.  Kernels with short hidden loops call begin()
    // early -- the band nets' at kernel start, the merger's before the band softmax -- so that the first fragments'
    // L2 round trip (~1.6 K cycles, exposed once per loop) travels behind other work.
    f4 bias0;
    __device__ __forceinline__ void begin(int ht0)
    {
        const int g = lane >> 4;
        bias0 = *reinterpret_cast<const f4 *>(b1 + 16 * min(ht0, hlast) + 4 * g);
#pragma unroll
        for (int k = 0; k < R; k++) {
            const int e = NOT + k;
            request(e % R, e % FP, ht0 - 1 + e / FP);
        }
        LCRC_FENCE();
    }
    __device__ __forceinline__ void finish(f4 (&acc)[NOT][FT], int ht0, int ht1)
    {
        if constexpr (BKQ > 0) {                  // the image is complete when finish() is called (begin() may run earlier)
#pragma unroll
            for (int kq = 0; kq < BKQ; kq++)
#pragma unroll
                for (int f = 0; f < FT; f++) bimg[kq][f] = XF[(f * NKQ + kq) * 64 + lane];
        }
        f4 bias = bias0;
        f4 pre[FT];
        pass<PRO>(acc, pre, bias, ht0 - 1);       // (a wave without tiles computes a dummy)
        for (int ht = ht0; ht < ht1 - 1; ht++) pass<MID>(acc, pre, bias, ht);
        if (ht0 < ht1) pass<LAST>(acc, pre, bias, ht1 - 1);
    }
    __device__ __forceinline__ void run(f4 (&acc)[NOT][FT], int ht0, int ht1)
    {
        begin(ht0);
        finish(acc, ht0, ht1);
    }
    // fields of a net's loop on this wave
    __device__ __forceinline__ void setup(const NetDev &nd, const f4 *xf_image, int lane_)
    {
        w1 = reinterpret_cast<const f4 *>(nd.w1p); w2 = reinterpret_cast<const f4 *>(nd.w2p);
        b1 = nd.b1; XF = xf_image; lane = lane_;
        hlast = nd.nht - 1;
        ks = nd.ksteps; nkq = nd.nkq; n_ot = nd.n_ot;
        z1 = w1 + (size_t)nd.nht * nd.nkq * 64; z2 = w2 + (size_t)nd.nht * nd.n_ot * 64;
        zi1 = nd.nht * nd.nkq; zi2 = nd.nht * nd.n_ot;
    }
};

// ---- split-f16 arithmetic (lcrc_set_arithmetic(h, LCRC_ARITH_SPLIT_F16)) -------------------------------------------
// An f32 product a*b is evaluated as three f16 MFMA products with f32 accumulation:
//     a = ah + al,  b = bh + bl   (ah = f16(a), al = f16(a - ah): 22 significant bits),   a*b ~ al*bh + ah*bl + ah*bh
// Every f16 x f16 product is exact in f32, so what is lost is the operands' 2^-22 tails and al*bl (2^-22 relative): the
// size of f32's own rounding of the running sum.  (tools/ubench/split_f16.hip: a two-layer product lands as close to
// f64 as v_mfma_f32_16x16x4_f32 does; f16 subnormals are not flushed by the MFMA.)  v_mfma_f32_16x16x32_f16 retires
// 16x the FLOPs per cycle of the f32 MFMA, so a product costs 3/16 of it -- and, unlike the f32 MFMA, it leaves the VALU
// free for half of its cycles.  Weights are split on the host (pack_net_h2), inputs when they are written into the
// operand image, hidden activations in registers.  Operands beyond +-65504 do not exist in f16: the host refuses models
// with such weights, inputs are clamped there (a normalised feature of that size saturates every sigmoid anyway).
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
constexpr float kHalfMax = 65504.0f;
// Operand scaling.  The low half of a value below 2^-3 is an f16 SUBNORMAL (absolute error 2^-25 instead of 2^-22
// relative), so every operand is moved up by a power of two before it is split -- exact, and undone exactly:
//   weights      by 2^e per matrix, max|w| * 2^e in (2^13, 2^14]   (host, pack_net_h2; the biases by the same factors)
//   net inputs   by kH2InScale  (normalised features are O(1..10); they clamp at +-65504 / kH2InScale = +-1023)
//   activations  by kH2ActScale (sigmoid outputs lie in (0, 1))
// layer-1 accumulators then hold 2^(e1 + 6) x the pre-activation -- undone inside FEXP's f64 product (NetDev::h2_sig_descale)
// --, layer-2 accumulators 2^(e2 + 14) x theirs -- undone when a wave publishes its partial tile (NetDev::h2_out_descale).
constexpr float kH2InScale = 64.0f, kH2ActScale = 16384.0f;

__device__ __forceinline__ f4 mfma_h(const f4 &a, const f4 &b, f4 c)
{
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, a), __builtin_bit_cast(h8, b), c, 0, 0, 0);
}

// Operand image of a net's input: [frame tile f][piece][k-step s][lane] 16-byte fragments; element j of lane l's
// fragment is X[frame 16f + (l&15)][k = 32s + 8(l>>4) + j], piece 0 = high part, 1 = low part.  The byte offset of
// (frame i, input k) is linear in k >> 3:  (i>>4) * 2*ns*1024 + (i&15) * 16  +  h2_k_ofs(k);  the low part lies
// ns * 1024 bytes further on.
__device__ __forceinline__ int h2_k_ofs(int k) { return ((k >> 3) << 8) + ((k & 7) << 1); }
__device__ __forceinline__ void h2_img_store(void *img, int ofs, int lo_ofs, float v)
{
    v = __builtin_amdgcn_fmed3f(v * kH2InScale, -kHalfMax, kHalfMax);
    const _Float16 hi = (_Float16)v;
    const _Float16 lo = (_Float16)(v - (float)hi);
    *reinterpret_cast<_Float16 *>(static_cast<char *>(img) + ofs) = hi;
    *reinterpret_cast<_Float16 *>(static_cast<char *>(img) + ofs + lo_ofs) = lo;
}

// Ring depth: what bounds the split-f16 loops at 32 frames per workgroup is the CU's vector-memory path (64 B/clk: every
// wave streams 42-54 KB of fragments per tile pair, 83 B/clk would be needed to keep up with the MFMAs), not the latency
// of the loads -- 8, 11/12 and 16 entries run alike (A/B run 17), so the shallowest ring (fewest registers) it is.
// For the same reason the sigmoid stays one block in front of a pass's MFMAs as in the f32 loop: dealt out over the
// layer-1 MFMAs' issue gaps (which the f16 MFMA, unlike the f32 one, leaves to the VALU) it gained nothing (A/B run 18).
constexpr int h2_ring_size()
{
    return 8;
}

// Hidden loop on tile PAIRS (32 hidden units): the layer-1 results of two 16-row tiles are, lane for lane, the B
// operand of one 32-deep layer-2 step -- k-slot 8g + j of lane (g, c) is hidden unit 32P + (j < 4 ? 4g + j : 16 + 4g + j - 4),
// which is how pack_net_h2 orders W2's columns.  Same ring discipline as RingLoop; per pair P the entries are
//     e <  NOT            W2(P)[e]            3*FT MFMAs (low x high, high x low, high x high)
//     e <  NOT + 2*NS     W1(P+1)[s][T]       s = (e-NOT)/2 k-step, T = (e-NOT)%2 tile of the pair
template <int KS, int NOT, int FT>
struct HalfLoop {
    static constexpr int NS = (4 * KS + 31) / 32;
    static constexpr int F = NOT + 2 * NS;
    static constexpr int R = h2_ring_size();
    static constexpr int FP = (F + R - 1) / R * R;
    enum { PRO = 0, MID = 1, LAST = 2 };

    const f4 *w1, *w2;
    const float *b1;
    const f4 *XF;
    int plast, lane;
    double sig_mul;             // FEXP's constant x 2^-(e1 + 6)
    f4 ring[R][2];

    __device__ __forceinline__ void request(int slot, int e, int P)
    {
        if (e < NOT) {
            gf4 *sp = scalar_ptr(w2 + ((size_t)max(0, min(P, plast)) * NOT + (e & ~1)) * 128);
            ring[slot][0] = sp[(e & 1) * 128 + lane];
            ring[slot][1] = sp[(e & 1) * 128 + 64 + lane];
        } else if (e - NOT < 2 * NS) {
            const int q = e - NOT;
            gf4 *sp = scalar_ptr(w1 + ((size_t)min(P + 1, plast) * (2 * NS) + (q & ~1)) * 128);
            ring[slot][0] = sp[(q & 1) * 128 + lane];
            ring[slot][1] = sp[(q & 1) * 128 + 64 + lane];
        }
    }

    template <int MODE>
    __device__ __forceinline__ void pass(f4 (&acc)[NOT][FT], f4 (&pre)[2 * FT], f4 (&bias)[2], int P)
    {
        const int g = lane >> 4;
        f4 sh[FT], sl[FT], nxt[2 * FT];
        if (MODE != PRO) {
            SigTile<2 * FT> sg;
            sg.begin(pre);
#pragma unroll
            for (int k = 0; k < SigTile<2 * FT>::kStages; k++) sg.stage(k, sig_mul);
            f4 s[2 * FT];
            sg.finish(s);
#pragma unroll
            for (int f = 0; f < FT; f++) {
                h8 hi, lo;
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    const float v = s[(j >> 2) * FT + f][j & 3] * kH2ActScale;
                    const _Float16 a = (_Float16)v;
                    hi[j] = a;
                    lo[j] = (_Float16)(v - (float)a);
                }
                sh[f] = __builtin_bit_cast(f4, hi);
                sl[f] = __builtin_bit_cast(f4, lo);
            }
        }
#pragma unroll
        for (int t = 0; t < 2; t++)
#pragma unroll
            for (int f = 0; f < FT; f++) nxt[t * FT + f] = bias[t];
        __builtin_amdgcn_sched_barrier(0);
        if (MODE != LAST) {
#pragma unroll
            for (int t = 0; t < 2; t++)
                bias[t] = *reinterpret_cast<const f4 *>(b1 + 32 * min(P + 2, plast) + 16 * t + 4 * g);
        }
        f4 xb[2][FT][2];
        if (MODE != LAST) {
#pragma unroll
            for (int f = 0; f < FT; f++)
#pragma unroll
                for (int pc = 0; pc < 2; pc++) xb[0][f][pc] = XF[((f * 2 + pc) * NS + 0) * 64 + lane];
        }
#pragma unroll
        for (int i = (MODE == PRO ? NOT : 0); i < (MODE == LAST ? NOT : FP); i++) {
            if (i < NOT) {
#pragma unroll
                for (int f = 0; f < FT; f++) {
                    acc[i][f] = mfma_h(ring[i % R][1], sh[f], acc[i][f]);
                    acc[i][f] = mfma_h(ring[i % R][0], sl[f], acc[i][f]);
                    acc[i][f] = mfma_h(ring[i % R][0], sh[f], acc[i][f]);
                }
            } else if (i - NOT < 2 * NS) {
                const int q = i - NOT, s = q >> 1, t = q & 1;
                if (t == 0 && s + 1 < NS) {
#pragma unroll
                    for (int f = 0; f < FT; f++)
#pragma unroll
                        for (int pc = 0; pc < 2; pc++) xb[(s + 1) & 1][f][pc] = XF[((f * 2 + pc) * NS + s + 1) * 64 + lane];
                }
#pragma unroll
                for (int f = 0; f < FT; f++) {
                    nxt[t * FT + f] = mfma_h(ring[i % R][1], xb[s & 1][f][0], nxt[t * FT + f]);
                    nxt[t * FT + f] = mfma_h(ring[i % R][0], xb[s & 1][f][1], nxt[t * FT + f]);
                    nxt[t * FT + f] = mfma_h(ring[i % R][0], xb[s & 1][f][0], nxt[t * FT + f]);
                }
            }
            const int e = (i + R) % FP, dp = (i + R) / FP;
            if ((MODE != LAST || (dp == 0 && e < NOT))) request(i % R, e, P + dp);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (MODE != LAST) {
#pragma unroll
            for (int q = 0; q < 2 * FT; q++) pre[q] = nxt[q];
        }
        LCRC_FENCE();
    }

    f4 bias0[2];
    __device__ __forceinline__ void begin(int p0)
    {
        const int g = lane >> 4;
#pragma unroll
        for (int t = 0; t < 2; t++) bias0[t] = *reinterpret_cast<const f4 *>(b1 + 32 * min(p0, plast) + 16 * t + 4 * g);
#pragma unroll
        for (int k = 0; k < R; k++) {
            const int e = NOT + k;
            request(e % R, e % FP, p0 - 1 + e / FP);
        }
        LCRC_FENCE();
    }
    __device__ __forceinline__ void finish(f4 (&acc)[NOT][FT], int p0, int p1)
    {
        f4 bias[2] = {bias0[0], bias0[1]};
        f4 pre[2 * FT];
        pass<PRO>(acc, pre, bias, p0 - 1);
        for (int pp = p0; pp < p1 - 1; pp++) pass<MID>(acc, pre, bias, pp);
        if (p0 < p1) pass<LAST>(acc, pre, bias, p1 - 1);
    }
    __device__ __forceinline__ void run(f4 (&acc)[NOT][FT], int p0, int p1)
    {
        begin(p0);
        finish(acc, p0, p1);
    }
    __device__ __forceinline__ void setup(const NetDev &nd, const f4 *xf_image, int lane_)
    {
        w1 = reinterpret_cast<const f4 *>(nd.w1h); w2 = reinterpret_cast<const f4 *>(nd.w2h);
        b1 = nd.b1h; XF = xf_image; lane = lane_;
        sig_mul = SigTile<1>::kFexpA * (double)nd.h2_sig_descale;
        plast = nd.npairs - 1;
    }
};

struct NoHook {
    __device__ __forceinline__ void operator()() const {}
};

// Layer-2 partial sums of ONE wave over the hidden tiles [ht0, ht1) of net `nd` (an empty range gives the
// bias or zeros): acc[ot][f][rr] = O^T[16ot + 4g + rr][16f + (lane&15)], started from the output bias
// (PrepareBiases nn.cpp:857) when with_b2.
template <int KS, int NOT, bool EXACT, int FT>
__device__ __forceinline__ void hidden_range(const NetDev &nd, const f4 *XF, int ht0, int ht1, bool with_b2, int lane,
                                             f4 (&acc)[NOT][FT])
{
    const int n_ot = EXACT ? NOT : nd.n_ot;
    const int g = lane >> 4;
#pragma unroll
    for (int ot = 0; ot < NOT; ot++) {
        f4 b = {0.f, 0.f, 0.f, 0.f};
        if (with_b2 && (EXACT || ot < n_ot))
            b = *reinterpret_cast<const f4 *>(nd.b2 + 16 * ot + 4 * g);
#pragma unroll
        for (int f = 0; f < FT; f++) acc[ot][f] = b;
    }
    RingLoop<KS, NOT, FT, EXACT> loop;
    // pointers into locals: kernarg fields would be re-read behind every memory fence
    loop.w1 = reinterpret_cast<const f4 *>(nd.w1p);
    loop.w2 = reinterpret_cast<const f4 *>(nd.w2p);
    loop.b1 = nd.b1; loop.XF = XF; loop.lane = lane;
    loop.hlast = nd.nht - 1;
    loop.ks = nd.ksteps; loop.nkq = nd.nkq; loop.n_ot = nd.n_ot;
    loop.z1 = loop.w1 + (size_t)nd.nht * nd.nkq * 64; loop.z2 = loop.w2 + (size_t)nd.nht * nd.n_ot * 64;
    loop.zi1 = nd.nht * nd.nkq; loop.zi2 = nd.nht * nd.n_ot;
    loop.run(acc, ht0, ht1);
}

// One wave's partial tile into its LDS slab ([ot][f][64] float4)
template <int NOT, bool EXACT, int FT>
__device__ __forceinline__ void store_partial(f4 *slab, int n_ot, int lane, const f4 (&acc)[NOT][FT])
{
    f4 *s = slab + lane;
#pragma unroll
    for (int ot = 0; ot < NOT; ot++)
        if (EXACT || ot < n_ot) {
#pragma unroll
            for (int f = 0; f < FT; f++) s[(ot * FT + f) * 64] = acc[ot][f];
        }
}

// Epilogues receive a lane's NV posteriors of one (net, frame) row at once: q[j] belongs to output
// o = part + LPF * j (valid while o < O; pad outputs carry 0), so that an epilogue can batch its LDS reads and
// transcendental work over the values instead of running one basic block per value.  per_value() adapts a simple
// per-value function `f(group, frame, o, posterior, valid)`.
template <typename F>
struct PerValueEpilogue {
    F f;
    int ovalid;      // outputs below it are valid whatever O is (0 = unknown)
    template <int LPF, int NV>
    __device__ __forceinline__ void operator()(int rg, int frame, int part, IntTag<LPF>, const float (&q)[NV], int O) const
    {
#pragma unroll
        for (int j = 0; j < NV; j++) f(rg, frame, part + LPF * j, q[j], LPF * j + LPF <= ovalid || part + LPF * j < O);
    }
};
template <typename F>
__device__ __forceinline__ PerValueEpilogue<F> per_value(F f, int ovalid = 0) { return PerValueEpilogue<F>{f, ovalid}; }

// Softmax (nn.cpp:822-855) in registers on all threads over output tiles that lie in LDS slabs: group g's
// pre-activations are the sum of its NPART partial slabs P[g][0..NPART) -- 1: as is, 2: p0 + p1,
// 4: (p0 + p1) + (p2 + p3), a fixed order.  LPF lanes share a (net, frame) row, each holds every LPF-th
// output.  Element (o, frame) of a slab: o = 16ot + 4g + rr, frame = 16f + c  ->  float index
// ((FT*ot + f)*64 + 16g + c)*4 + rr.  Ends like run_net (epi called, one __syncthreads() before the calls).
// (the slab pointers are passed one by one and group 1's are selected with ?: -- through an array of
//  pointers hipcc loses the LDS address space and reads the slabs with flat loads)
// OVALID: outputs below it are valid whatever n_out is (exact variants: n_ot == NOT, so n_out > 16 * (NOT - 1)); their
// pad selects disappear at compile time.
template <int NOT, int NW, int FT, int GROUPS, int NPART, int OVALID, typename Params, typename Epi>
__device__ __forceinline__ void softmax_rows(const Params &prm, const NetDev *nets, const float *a0, const float *a1,
                                             const float *a2, const float *a3, const float *b0, const float *b1,
                                             int lane, int wave, Epi epi)
{
    constexpr int BM = 16 * FT;
    constexpr int LPF = NW * 64 / (GROUPS * BM);   // lanes cooperating on one row
    constexpr int NV = 16 * NOT / LPF;             // values per lane
    static_assert(LPF == 4 || LPF == 8 || LPF == 16, "softmax lane groups are 4, 8 or 16 wide");
    static_assert(NPART == 1 || NPART == 2 || NPART == 4, "1, 2 or 4 partial slabs");
    const int tid = wave * 64 + lane;
    const int row = tid / LPF, part = tid % LPF;
    const int rg = GROUPS == 1 ? 0 : row / BM;     // wave-uniform: a wave's rows belong to one net
    const int frame = GROUPS == 1 ? row : row % BM;
    static_assert(GROUPS == 1 || NPART <= 2, "two groups: one or two partial slabs each");
    const float *sa = rg == 0 ? a0 : b0, *sb = rg == 0 ? a1 : b1;
    const float *sc = a2, *sd = a3;
    // o = part + LPF*j: because LPF is a multiple of 4 and part < LPF <= 16, the slab index of o splits
    // into a per-thread part and a COMPILE-TIME part of j (no carries between the bit fields), so the
    // reads below are base + immediate offset
    const int pbase = ((frame >> 4) * 64 + (frame & 15)) * 4 + ((part >> 2) & 3) * 64 + (part & 3);
    const int O = nets[rg].n_out;
    float v[NV];
    float m = -FLT_MAX;
#pragma unroll
    for (int j = 0; j < NV; j++) {
        const int o = part + LPF * j;
        // o < 16*NOT always addresses the slab (pad outputs hold zero weights' sums): read
        // unconditionally so the LDS reads are issued back to back, select afterwards
        const int idx = pbase + ((LPF * j) >> 4) * (256 * FT) + (((LPF * j) >> 2) & 3) * 64;
        float t = sa[idx];
        if (NPART >= 2) t += sb[idx];
        if (NPART == 4) t += sc[idx] + sd[idx];        // waves (0 + 1) + (2 + 3): a fixed order
        v[j] = (LPF * j + LPF <= OVALID || o < O) ? t : -FLT_MAX;
        m = fmaxf(m, v[j]);
    }
    m = allreduce<LPF>(m, [](float a, float b) { return fmaxf(a, b); });
    // The sum is grouped as 16 strided partials (o mod 16) combined by a fixed butterfly (pairs, then
    // quads, then halves of 8, then the two halves), whatever LPF is: with fewer than 16 lanes per row a
    // lane carries 16 / LPF of the partials and the last butterfly steps become plain adds -- so every
    // geometry (16- or 32-frame workgroups, one or two nets at a time) produces the same bits.
    constexpr int PPL = 16 / LPF;                  // partials per lane
    float ps[PPL];
#pragma unroll
    for (int q = 0; q < PPL; q++) ps[q] = 0.0f;
#pragma unroll
    for (int j = 0; j < NV; j++) {
        const float e = fexp_nonpos_f(v[j] - m);     // pads: FEXP(-FLT_MAX - m) is computed and discarded
        v[j] = (LPF * j + LPF <= OVALID || part + LPF * j < O) ? e : 0.0f;
        ps[j % PPL] += v[j];
    }
#pragma unroll
    for (int q = 0; q < PPL; q++) ps[q] = allreduce<LPF>(ps[q], [](float a, float b) { return a + b; });
    float sum;
    if constexpr (PPL == 1) sum = ps[0];
    else if constexpr (PPL == 2) sum = ps[0] + ps[1];
    else sum = (ps[0] + ps[1]) + (ps[2] + ps[3]);
    const float scale = 1.0f / sum;
    __syncthreads();                          // slabs are free again (the epilogue may reuse them)
    LCRC_STAMP(prm, wave, lane, 13);
#pragma unroll
    for (int j = 0; j < NV; j++) v[j] *= scale;
    epi(rg, frame, part, IntTag<LPF>(), v, O);    // values of pad outputs (o >= O) are 0: loads only, no store
}

// Runs GROUPS nets of the same shape class at once, each on NW / GROUPS waves (GROUPS = 1: one net on all
// waves; GROUPS = 2: the two band classifiers of LCRC side by side on wave pairs -- one hidden loop, one
// softmax phase and no fold round instead of two of each).  nets[g] / XF + g * xf_stride belong to group g.
// slab01: two slabs of FT * n_ot_slab KiB; slab23: two more (with GROUPS = 2 they may alias the nets' B
// images and anything else that is dead once every wave has left its hidden loop; with GROUPS = 1 they must
// be free when the first wave leaves its loop).
// On return the row epilogue `epi(group, frame, part, IntTag<LPF>, q[NV], n_out)` (see PerValueEpilogue) has
// been called by every thread: each (group, frame, output < n_out) is covered once by SOME thread; pad
// outputs o in [n_out, 16 * n_ot) may be used for loads from arrays padded to the output tiles but must not
// be stored.  A __syncthreads() has been passed.
// EARLY: `early` is this wave's loop, set up and begun by the caller (RingLoop::begin) -- its first fragments are already
// travelling.  hook() runs right after the partial tiles are published, before the softmax (a place to begin() the
// NEXT net's loop).
template <int KS, int NOT, int NW, bool EXACT, int FT, int GROUPS, bool EARLY = false, int BKQ = 0, int ARITH = 0,
          typename Params, typename Epi, typename Hook = NoHook>
__device__ __forceinline__ void run_net(const Params &prm, int stamp0, const NetDev *nets,
                                        const f4 *__restrict__ XFbase, int xf_stride, f4 *__restrict__ slab01,
                                        f4 *__restrict__ slab23, int n_ot_slab, int lane, int wave, Epi epi,
                                        RingLoop<KS, NOT, FT, EXACT, BKQ> *early = nullptr, Hook hook = Hook())
{
    constexpr int WPG = NW / GROUPS;             // waves per net
    const int grp = GROUPS == 1 ? 0 : wave / WPG, wig = GROUPS == 1 ? wave : wave % WPG;
    const NetDev &nd = nets[grp];
    const int n_ot = EXACT ? NOT : nd.n_ot;
    const int slab_f4 = FT * n_ot_slab * 64;    // float4 per slab
    const f4 *XF = XFbase + (size_t)grp * xf_stride;
    const int g = lane >> 4;

    // (this is hidden_range() + store_partial(), spelled out: as calls they compile to the same arithmetic
    //  but hipcc's register allocation of the 32-frame variants gets worse -- 57 instead of 1 AGPR copies
    //  for cz_42_69_9, accvgpr moves inside the merger's loop, 2 % slower in same-GPU A/B runs)
    // layer-2 accumulators: acc[ot][f][rr] = O^T[16ot + 4g + rr][16f + (lane&15)]
    f4 acc[NOT][FT];
#pragma unroll
    for (int ot = 0; ot < NOT; ot++) {
        f4 b = {0.f, 0.f, 0.f, 0.f};
        if (wig == 0 && (EXACT || ot < n_ot))
            b = *reinterpret_cast<const f4 *>((ARITH ? nd.b2h : nd.b2) + 16 * ot + 4 * g);   // PrepareBiases nn.cpp:857
#pragma unroll
        for (int f = 0; f < FT; f++) acc[ot][f] = b;
    }

    static_assert(ARITH == 0 || (EXACT && !EARLY), "split-f16 arithmetic: compile-time shapes");
    const int units = ARITH ? nd.npairs : nd.nht;        // what the waves share: hidden tiles, or tile pairs
    const int tpw = (units + WPG - 1) / WPG;
    const int ht0 = wig * tpw;
    const int ht1 = min(units, ht0 + tpw);
    // pointers into locals: kernarg fields would be re-read behind every memory fence
    const f4 *const w1 = reinterpret_cast<const f4 *>(nd.w1p);
    const f4 *const w2 = reinterpret_cast<const f4 *>(nd.w2p);
    const float *const b1 = nd.b1;
    const int hlast = nd.nht - 1;

    if constexpr (ARITH == 1) {
        HalfLoop<KS, NOT, FT> loop;
        loop.setup(nd, XF, lane);
        loop.run(acc, ht0, ht1);
    } else if constexpr (EARLY) {
        early->finish(acc, ht0, ht1);
    } else {
        RingLoop<KS, NOT, FT, EXACT> loop;
        loop.w1 = w1; loop.w2 = w2; loop.b1 = b1; loop.XF = XF; loop.hlast = hlast; loop.lane = lane;
        loop.ks = nd.ksteps; loop.nkq = nd.nkq; loop.n_ot = nd.n_ot;
        loop.z1 = w1 + (size_t)nd.nht * nd.nkq * 64; loop.z2 = w2 + (size_t)nd.nht * nd.n_ot * 64;
        loop.zi1 = nd.nht * nd.nkq; loop.zi2 = nd.nht * nd.n_ot;
        loop.run(acc, ht0, ht1);
    }

    LCRC_STAMP(prm, wave, lane, stamp0);       // hidden loop done
    if constexpr (ARITH == 1) {                // the accumulators hold 2^(e2 + 14) x the sums (exact to undo)
        const float ds = nd.h2_out_descale;
#pragma unroll
        for (int ot = 0; ot < NOT; ot++)
#pragma unroll
            for (int f = 0; f < FT; f++) acc[ot][f] *= ds;
    }
    {
        // No fold round: every wave publishes its partial tile in a slab of its own and the softmax adds
        // them while it reads (two per net when two nets share the waves, four otherwise).  slab23 may lie over
        // the B images: wait until every wave has left its hidden loop before writing there.
        if (GROUPS == 2) __syncthreads();
        f4 *s = (GROUPS == 2 ? (grp == 0 ? slab01 : slab23) + wig * slab_f4
                             : (wig < 2 ? slab01 : slab23) + (wig & 1) * slab_f4) + lane;
#pragma unroll
        for (int ot = 0; ot < NOT; ot++)
            if (EXACT || ot < n_ot) {
#pragma unroll
                for (int f = 0; f < FT; f++) s[(ot * FT + f) * 64] = acc[ot][f];
            }
        __syncthreads();
    }
    hook();
    LCRC_STAMP(prm, wave, lane, 12);           // partial tiles published (last net's value survives)
    const float *s01 = reinterpret_cast<const float *>(slab01), *s23 = reinterpret_cast<const float *>(slab23);
    constexpr int OVALID = EXACT ? 16 * (NOT - 1) : 0;
    if constexpr (GROUPS == 2)
        softmax_rows<NOT, NW, FT, 2, 2, OVALID>(prm, nets, s01, s01 + slab_f4 * 4, s01, s01, s23, s23 + slab_f4 * 4, lane, wave, epi);
    else
        softmax_rows<NOT, NW, FT, 1, 4, OVALID>(prm, nets, s01, s01 + slab_f4 * 4, s23, s23 + slab_f4 * 4, s01, s01, lane, wave, epi);
    __syncthreads();
}

}  // namespace phnrec
#endif
