// phndec.cpp -- see phndec.h
#include "phndec.h"

#include <immintrin.h>

#include <cfloat>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>

namespace phnrec {

namespace {
const float kLogHalf = -0.69314718055994530941723212145818f;   // both transitions (phndec.cpp:9,14-15)
bool UseAvx512();
}

bool PhnDec::LoadPhnList(const std::string &path)
{
    std::ifstream in(path.c_str());
    if (!in) return false;
    phn_.clear();
    std::string line;
    while (std::getline(in, line)) {
        size_t e = line.find_first_of("\r\n");
        if (e != std::string::npos) line.erase(e);
        phn_.push_back(line);
    }
    return true;
}

namespace { constexpr int kPackMaxPhonemes = 127; }

void PhnDec::Init()
{
    const int P = (int)phn_.size(), W = S_ + 1;
    Pp_ = (P + 15) & ~15;
    alpha_.assign((size_t)W * Pp_, -FLT_MAX);
    prev_.assign((size_t)W * Pp_, -1);
    len_.assign((size_t)W * Pp_, 0);
    obs_.assign((size_t)S_ * Pp_, 0.0f);
    hphn_.assign(prune_ + 1, -1);
    hlen_.assign(prune_ + 1, -1);
    halpha_.assign(prune_ + 1, -1.0f);
    hpos_ = 0;
    for (int i = 0; i < P; i++) alpha_[i] = wpen_;               // entry state carries the penalty
    // (the packed form carries (winner + 1) << 24 | length in a signed word: at most 127 phonemes; beyond that the AVX2
    //  form, which like the plain one and the reference has no limit)
    packed_ = S_ == 3 && P <= kPackMaxPhonemes && UseAvx512();
    if (packed_) pk_.assign((size_t)W * Pp_, 0);                 // (winner -1, length 0) everywhere
    entry_a_ = wpen_;
    entry_prev_ = -1;
    nframes_ = 0;
    prev_alpha_ = 0.0f;
    labels_.clear();
}

namespace {

// The per-frame work below exists twice: plain C++ and AVX2 (chosen once per process from the CPU's feature bits).
// Both perform the SAME IEEE additions and comparisons on the same values -- the vector form only does eight
// phonemes at a time -- so labels, times and scores do not depend on which one runs (tests/test_cli_cpu.py decodes
// the reference's posterior dumps with both, PHNREC_NO_AVX2=1 forcing the plain one).
bool UseAvx2()
{
    static const bool v = __builtin_cpu_supports("avx2") && !getenv("PHNREC_NO_AVX2");
    return v;
}
// ... and a third time on AVX-512 (sixteen phonemes per instruction, the whole frame in one pass over the token slots):
// PHNREC_NO_AVX512=1 keeps to AVX2, PHNREC_NO_AVX2=1 to the plain form.  Still the same IEEE operations.
bool UseAvx512()
{
    static const bool v = UseAvx2() && __builtin_cpu_supports("avx512f") && !getenv("PHNREC_NO_AVX512");
    return v;
}

// One state of every phoneme (phndec.cpp:160-167, the inner loop's body): the token of state j stays, or is replaced
// by the one of state j - 1.
void UpdateStatePlain(int n, float *a_j, const float *a_jm1, int *pv_j, const int *pv_jm1, int *ln_j, const int *ln_jm1,
                      const float *obs)
{
    for (int i = 0; i < n; i++) {
        const float stay = a_j[i] + kLogHalf, enter = a_jm1[i] + kLogHalf;
        if (stay > enter) {
            a_j[i] = stay + obs[i];
            ln_j[i] += 1;
        } else {
            a_j[i] = enter + obs[i];
            pv_j[i] = pv_jm1[i];
            ln_j[i] = ln_jm1[i] + 1;
        }
    }
}

__attribute__((target("avx2")))
void UpdateStateAvx2(int n, float *a_j, const float *a_jm1, int *pv_j, const int *pv_jm1, int *ln_j, const int *ln_jm1,
                     const float *obs)
{
    const __m256 c = _mm256_set1_ps(kLogHalf);
    const __m256i one = _mm256_set1_epi32(1);
    for (int i = 0; i < n; i += 8) {
        const __m256 stay = _mm256_add_ps(_mm256_loadu_ps(a_j + i), c), enter = _mm256_add_ps(_mm256_loadu_ps(a_jm1 + i), c);
        const __m256 keep = _mm256_cmp_ps(stay, enter, _CMP_GT_OQ);
        _mm256_storeu_ps(a_j + i, _mm256_add_ps(_mm256_blendv_ps(enter, stay, keep), _mm256_loadu_ps(obs + i)));
        const __m256i k = _mm256_castps_si256(keep);
        const __m256i pv = _mm256_blendv_epi8(_mm256_loadu_si256((const __m256i *)(pv_jm1 + i)),
                                              _mm256_loadu_si256((const __m256i *)(pv_j + i)), k);
        const __m256i ln = _mm256_blendv_epi8(_mm256_loadu_si256((const __m256i *)(ln_jm1 + i)),
                                              _mm256_loadu_si256((const __m256i *)(ln_j + i)), k);
        _mm256_storeu_si256((__m256i *)(pv_j + i), pv);
        _mm256_storeu_si256((__m256i *)(ln_j + i), _mm256_add_epi32(ln, one));
    }
}

// A frame's log-posteriors ([phoneme][state], pdf of state j of phoneme i at i*S + j) state-major into obs[j][Pp]
void TransposePlain(const float *f, int P, int S, int Pp, float *obs, int first = 0)
{
    for (int i = first; i < P; i++)
        for (int j = 0; j < S; j++) obs[(size_t)j * Pp + i] = f[i * S + j];
}

// ... for three states per phoneme (every shipped system): 24 consecutive values are eight phonemes; element p of the
// blend below comes from the vector that holds a wanted value in lane p, the permutation puts them in phoneme order
__attribute__((target("avx2")))
void Transpose3Avx2(const float *f, int P, int Pp, float *obs)
{
    const __m256i p0 = _mm256_setr_epi32(0, 3, 6, 1, 4, 7, 2, 5), p1 = _mm256_setr_epi32(1, 4, 7, 2, 5, 0, 3, 6),
                  p2 = _mm256_setr_epi32(2, 5, 0, 3, 6, 1, 4, 7);
    int i = 0;
    for (; i + 8 <= P; i += 8) {
        const __m256 v0 = _mm256_loadu_ps(f + 3 * i), v1 = _mm256_loadu_ps(f + 3 * i + 8), v2 = _mm256_loadu_ps(f + 3 * i + 16);
        // state 0: values 0,3,6 | 9,12,15 | 18,21 -> lanes 0,3,6 of v0, 1,4,7 of v1, 2,5 of v2
        const __m256 s0 = _mm256_blend_ps(_mm256_blend_ps(v0, v1, 0x92), v2, 0x24);
        // state 1: values 1,4,7 | 10,13 | 16,19,22 -> lanes 1,4,7 of v0, 2,5 of v1, 0,3,6 of v2
        const __m256 s1 = _mm256_blend_ps(_mm256_blend_ps(v0, v1, 0x24), v2, 0x49);
        // state 2: values 2,5 | 8,11,14 | 17,20,23 -> lanes 2,5 of v0, 0,3,6 of v1, 1,4,7 of v2
        const __m256 s2 = _mm256_blend_ps(_mm256_blend_ps(v0, v1, 0x49), v2, 0x92);
        _mm256_storeu_ps(obs + i, _mm256_permutevar8x32_ps(s0, p0));
        _mm256_storeu_ps(obs + Pp + i, _mm256_permutevar8x32_ps(s1, p1));
        _mm256_storeu_ps(obs + 2 * (size_t)Pp + i, _mm256_permutevar8x32_ps(s2, p2));
    }
    TransposePlain(f, P, 3, Pp, obs, i);         // the last, partial group (never reads past the frame's row)
}

// Largest value of a[0..n) (n a multiple of 8; pads hold -FLT_MAX), and the first index that holds `v` (n if none)
__attribute__((target("avx2")))
float RowMaxAvx2(const float *a, int n)
{
    __m256 m = _mm256_loadu_ps(a);
    for (int i = 8; i < n; i += 8) m = _mm256_max_ps(m, _mm256_loadu_ps(a + i));
    __m128 h = _mm_max_ps(_mm256_castps256_ps128(m), _mm256_extractf128_ps(m, 1));
    h = _mm_max_ps(h, _mm_movehl_ps(h, h));
    h = _mm_max_ss(h, _mm_shuffle_ps(h, h, 1));
    return _mm_cvtss_f32(h);
}

__attribute__((target("avx2")))
int FirstEqualAvx2(const float *a, int n, float v)
{
    const __m256 vv = _mm256_set1_ps(v);
    for (int i = 0; i < n; i += 8) {
        const int mask = _mm256_movemask_ps(_mm256_cmp_ps(_mm256_loadu_ps(a + i), vv, _CMP_EQ_OQ));
        if (mask) return i + __builtin_ctz((unsigned)mask);
    }
    return n;
}

__attribute__((target("avx2")))
void FillEntryAvx2(int n, float *a, int *pv, int *ln, float entry, int bi)
{
    const __m256 e = _mm256_set1_ps(entry);
    const __m256i b = _mm256_set1_epi32(bi), z = _mm256_setzero_si256();
    for (int i = 0; i < n; i += 8) {
        _mm256_storeu_ps(a + i, e);
        _mm256_storeu_si256((__m256i *)(pv + i), b);
        _mm256_storeu_si256((__m256i *)(ln + i), z);
    }
}

// ---- AVX-512, three states per phoneme (every shipped system) --------------------------------------------------
// One pass per group of sixteen phonemes: the group's 48 log-posteriors are loaded (masked at the end of the phoneme
// list: never past what the list needs) and brought state-major by two permutes per state; the token slots of the
// sixteen models are loaded once, updated last state first on the OLD values exactly as phndec.cpp:96-119 does, and
// stored; the running maxima of the exit row and of all inner rows come out of the same pass (max is exact: any order).
// Two things are kept differently from the other forms, to halve the loads and stores of a frame:
//   * a token's (entry winner, length) pair travels as ONE word, (winner + 1) << 24 | length (kPackLenBits; the frame
//     that would overflow the length field unpacks the slots and the utterance goes on in the AVX2 form);
//   * the entry row is not stored at all: after PropagateInNetwork every phoneme's entry slot holds the same token
//     (score best + penalty, winner bi, length 0; phndec.cpp:121-144), so it is three scalars that are broadcast.
constexpr int kPackLenBits = 24;
static_assert(((long long)(kPackMaxPhonemes) << kPackLenBits) <= 0x7fffffffLL, "winner + 1 must fit above the length bits");
struct Frame512 {
    __m512 exit_max, inner_max;
};

__attribute__((target("avx512f")))
inline Frame512 UpdateFrame3Avx512(const float *f, int P, int Pp, float *alpha, int *pk, float entry, int entry_pk)
{
    const __m512 c = _mm512_set1_ps(kLogHalf);
    const __m512i one = _mm512_set1_epi32(1);
    // state s of phoneme p of the group is value 3p + s of its 48: values 0..31 come from (v0, v1), 32..47 from v2
    const __m512i i0 = _mm512_setr_epi32(0, 3, 6, 9, 12, 15, 18, 21, 24, 27, 30, 0, 0, 0, 0, 0);
    const __m512i i1 = _mm512_setr_epi32(1, 4, 7, 10, 13, 16, 19, 22, 25, 28, 31, 0, 0, 0, 0, 0);
    const __m512i i2 = _mm512_setr_epi32(2, 5, 8, 11, 14, 17, 20, 23, 26, 29, 0, 0, 0, 0, 0, 0);
    const __m512i t0 = _mm512_setr_epi32(0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 4, 7, 10, 13);     // value 33.. -> v2[1..]
    const __m512i t1 = _mm512_setr_epi32(0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 2, 5, 8, 11, 14);
    const __m512i t2 = _mm512_setr_epi32(0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 3, 6, 9, 12, 15);     // p = 10 is value 32 = v2[0]
    Frame512 r;
    r.exit_max = _mm512_set1_ps(-FLT_MAX);
    r.inner_max = _mm512_set1_ps(-FLT_MAX);
    float *a1 = alpha + Pp, *a2 = alpha + 2 * (size_t)Pp, *a3 = alpha + 3 * (size_t)Pp;
    int *p1 = pk + Pp, *p2 = pk + 2 * (size_t)Pp, *p3 = pk + 3 * (size_t)Pp;
    const __m512 ev = _mm512_set1_ps(entry), floor = _mm512_set1_ps(-FLT_MAX);
    const __m512i P0 = _mm512_set1_epi32(entry_pk);
    for (int i = 0; i < Pp; i += 16) {
        const int nv = 3 * (P - i);                                  // values of this group that exist (>= 48: all)
        const __mmask16 m0 = nv >= 16 ? 0xFFFF : (__mmask16)((1u << (nv > 0 ? nv : 0)) - 1);
        const __mmask16 m1 = nv >= 32 ? 0xFFFF : (__mmask16)((1u << (nv > 16 ? nv - 16 : 0)) - 1);
        const __mmask16 m2 = nv >= 48 ? 0xFFFF : (__mmask16)((1u << (nv > 32 ? nv - 32 : 0)) - 1);
        const __m512 v0 = _mm512_maskz_loadu_ps(m0, f + 3 * i), v1 = _mm512_maskz_loadu_ps(m1, f + 3 * i + 16),
                     v2 = _mm512_maskz_loadu_ps(m2, f + 3 * i + 32);
        // state 0: p = 0..10 at 3p <= 30 in (v0, v1); p = 11..15 at 33..45 = v2[1, 4, 7, 10, 13]
        const __m512 o0 = _mm512_mask_permutexvar_ps(_mm512_permutex2var_ps(v0, i0, v1), 0xF800, t0, v2);
        // state 1: p = 0..10 at 3p + 1 <= 31; p = 11..15 at 34..46 = v2[2, 5, 8, 11, 14]
        const __m512 o1 = _mm512_mask_permutexvar_ps(_mm512_permutex2var_ps(v0, i1, v1), 0xF800, t1, v2);
        // state 2: p = 0..9 at 3p + 2 <= 29; p = 10 at 32 = v2[0]; p = 11..15 at 35..47 = v2[3, 6, 9, 12, 15]
        const __m512 o2 = _mm512_mask_permutexvar_ps(_mm512_permutex2var_ps(v0, i2, v1), 0xFC00, t2, v2);

        const int np = P - i;                                        // phonemes of this group that exist
        const __mmask16 mp = np >= 16 ? 0xFFFF : (__mmask16)((1u << (np > 0 ? np : 0)) - 1);
        const __m512 A0 = _mm512_mask_blend_ps(mp, floor, ev);       // pad slots keep -FLT_MAX (they never win)
        const __m512 A1 = _mm512_loadu_ps(a1 + i), A2 = _mm512_loadu_ps(a2 + i), A3 = _mm512_loadu_ps(a3 + i);
        const __m512 e0 = _mm512_add_ps(A0, c), e1 = _mm512_add_ps(A1, c), e2 = _mm512_add_ps(A2, c), e3 = _mm512_add_ps(A3, c);
        // state 3: stays (e3) or takes state 2's token (e2); state 2: e2 vs e1; state 1: e1 vs e0 -- all on old values
        const __mmask16 k3 = _mm512_cmp_ps_mask(e3, e2, _CMP_GT_OQ), k2 = _mm512_cmp_ps_mask(e2, e1, _CMP_GT_OQ),
                        k1 = _mm512_cmp_ps_mask(e1, e0, _CMP_GT_OQ);
        const __m512 n3 = _mm512_add_ps(_mm512_mask_blend_ps(k3, e2, e3), o2);
        const __m512 n2 = _mm512_add_ps(_mm512_mask_blend_ps(k2, e1, e2), o1);
        const __m512 n1 = _mm512_add_ps(_mm512_mask_blend_ps(k1, e0, e1), o0);
        _mm512_storeu_ps(a3 + i, n3);
        _mm512_storeu_ps(a2 + i, n2);
        _mm512_storeu_ps(a1 + i, n1);
        const __m512i P1 = _mm512_loadu_si512(p1 + i), P2 = _mm512_loadu_si512(p2 + i), P3 = _mm512_loadu_si512(p3 + i);
        _mm512_storeu_si512(p3 + i, _mm512_add_epi32(_mm512_mask_blend_epi32(k3, P2, P3), one));      // length + 1
        _mm512_storeu_si512(p2 + i, _mm512_add_epi32(_mm512_mask_blend_epi32(k2, P1, P2), one));
        _mm512_storeu_si512(p1 + i, _mm512_add_epi32(_mm512_mask_blend_epi32(k1, P0, P1), one));
        r.exit_max = _mm512_max_ps(r.exit_max, n3);
        r.inner_max = _mm512_max_ps(r.inner_max, _mm512_max_ps(n3, _mm512_max_ps(n2, n1)));
    }
    return r;
}

__attribute__((target("avx512f")))
inline int FirstEqualAvx512(const float *a, int n, float v)
{
    const __m512 vv = _mm512_set1_ps(v);
    for (int i = 0; i < n; i += 16) {
        const unsigned mask = _mm512_cmp_ps_mask(_mm512_loadu_ps(a + i), vv, _CMP_EQ_OQ);
        if (mask) return i + __builtin_ctz(mask);
    }
    return n;
}

__attribute__((target("avx512f")))
inline float HMax512(__m512 v) { return _mm512_reduce_max_ps(v); }

}  // namespace

// packed form <-> the (prev_, len_) arrays of the other forms
void PhnDec::Unpack()
{
    const int P = (int)phn_.size(), Pp = Pp_, W = S_ + 1;
    for (int j = 1; j < W; j++)
        for (int i = 0; i < Pp; i++) {
            const int w = pk_[(size_t)j * Pp + i];
            prev_[(size_t)j * Pp + i] = (w >> kPackLenBits) - 1;
            len_[(size_t)j * Pp + i] = w & ((1 << kPackLenBits) - 1);
        }
    for (int i = 0; i < P; i++) { alpha_[i] = entry_a_; prev_[i] = entry_prev_; len_[i] = 0; }
    packed_ = false;
}

// the whole frame on AVX-512 (S == 3): the same steps as ProcessFrame below
__attribute__((target("avx512f")))
void PhnDec::ProcessFrame512(const float *f)
{
    const int P = (int)phn_.size(), Pp = Pp_;
    const Frame512 fm = UpdateFrame3Avx512(f, P, Pp, alpha_.data(), pk_.data(), entry_a_, (entry_prev_ + 1) << kPackLenBits);
    const float *exitv = &alpha_[(size_t)3 * Pp];
    float best = -FLT_MAX;
    int bi = 0;
    const float m = HMax512(fm.exit_max);
    if (m > best) { best = m; bi = FirstEqualAvx512(exitv, Pp, m); }
    const int cols = (int)hphn_.size();
    const int slot = hpos_;
    hpos_ = hpos_ + 1 == cols ? 0 : hpos_ + 1;
    const int w = pk_[(size_t)3 * Pp + bi];
    hphn_[slot] = (w >> kPackLenBits) - 1;
    hlen_[slot] = w & ((1 << kPackLenBits) - 1);
    halpha_[slot] = best;
    entry_a_ = best + wpen_;
    entry_prev_ = bi;
    nframes_++;
    if (nframes_ < cols) return;
    // TimePruning's best inner token (phndec.cpp:191-205): smallest phoneme index among equals, then smallest state
    int ti = -1, tj = 0;
    const float im = HMax512(fm.inner_max);
    if (im > -FLT_MAX) {
        ti = Pp;
        for (int j = 1; j <= 3; j++) {
            const int i = FirstEqualAvx512(&alpha_[(size_t)j * Pp], Pp, im);
            if (i < ti) { ti = i; tj = j; }
        }
    }
    PruneFrom(ti, tj);
}

void PhnDec::ProcessFrame(const float *f)
{
    if (packed_) {
        if (nframes_ < (1 << kPackLenBits) - 2) { ProcessFrame512(f); return; }
        Unpack();                            // an utterance of 2^24 frames: the length field is full, go on unpacked
    }
    const int P = (int)phn_.size(), Pp = Pp_;
    const bool avx2 = UseAvx2();
    // this frame's observations state-major
    if (avx2 && S_ == 3) Transpose3Avx2(f, P, Pp, obs_.data());
    else TransposePlain(f, P, S_, Pp, obs_.data());
    // inside the models, last state first
    for (int j = S_; j > 0; j--) {
        float *a = &alpha_[(size_t)j * Pp];
        int *pv = &prev_[(size_t)j * Pp], *ln = &len_[(size_t)j * Pp];
        if (avx2) UpdateStateAvx2(Pp, a, a - Pp, pv, pv - Pp, ln, ln - Pp, &obs_[(size_t)(j - 1) * Pp]);
        else UpdateStatePlain(P, a, a - Pp, pv, pv - Pp, ln, ln - Pp, &obs_[(size_t)(j - 1) * Pp]);
    }
    // network level: the best exit token re-enters every phoneme (first strict maximum)
    const float *exitv = &alpha_[(size_t)S_ * Pp];
    float best = -FLT_MAX;
    int bi = 0;
    if (avx2) {
        const float m = RowMaxAvx2(exitv, Pp);
        if (m > best) { best = m; bi = FirstEqualAvx2(exitv, Pp, m); }
    } else {
        for (int i = 0; i < P; i++)
            if (exitv[i] > best) { best = exitv[i]; bi = i; }
    }
    // the history's oldest column is overwritten: the ring advances by one
    const int cols = (int)hphn_.size();
    const int slot = hpos_;                  // column cols-1 after the advance
    hpos_ = hpos_ + 1 == cols ? 0 : hpos_ + 1;
    hphn_[slot] = prev_[(size_t)S_ * Pp + bi];
    hlen_[slot] = len_[(size_t)S_ * Pp + bi];
    halpha_[slot] = best;
    const float entry = best + wpen_;
    if (avx2) {
        // (pad slots of the entry row take the value too: they feed pad slots of state 1, which no search looks at --
        //  the row maxima above run over rows 1..S, whose pads would then no longer be -FLT_MAX; so only P are written)
        FillEntryAvx2(P & ~7, alpha_.data(), prev_.data(), len_.data(), entry, bi);
        for (int i = P & ~7; i < P; i++) { alpha_[i] = entry; prev_[i] = bi; len_[i] = 0; }
    } else {
        for (int i = 0; i < P; i++) { alpha_[i] = entry; prev_[i] = bi; len_[i] = 0; }
    }
    nframes_++;
    TimePruning();
}

// phndec.cpp:191-234
void PhnDec::TimePruning()
{
    const int cols = (int)hlen_.size(), P = (int)phn_.size(), Pp = Pp_;
    if (nframes_ < cols) return;
    // the best token inside the models: the reference scans phoneme-major (states 1..S inside) and keeps the first strict
    // maximum, i.e. among equal values the smallest phoneme index, then the smallest state
    float best = -FLT_MAX;
    int bi = -1, bj = 0;
    if (UseAvx2()) {
        float m = -FLT_MAX;
        for (int j = 1; j <= S_; j++) {
            const float r = RowMaxAvx2(&alpha_[(size_t)j * Pp], Pp);
            m = r > m ? r : m;
        }
        if (m > best) {
            best = m;
            bi = Pp;
            for (int j = 1; j <= S_; j++) {
                const int i = FirstEqualAvx2(&alpha_[(size_t)j * Pp], Pp, m);
                if (i < bi) { bi = i; bj = j; }
            }
        }
    } else {
        for (int j = 1; j <= S_; j++) {
            const float *a = &alpha_[(size_t)j * Pp];
            for (int i = 0; i < P; i++)
                if (a[i] > best || (a[i] == best && i < bi)) { best = a[i]; bi = i; bj = j; }
        }
    }
    PruneFrom(bi, bj);
}

// the walk from the best inner token (state bj of phoneme bi; bi < 0: none beat -FLT_MAX) back to the horizon
void PhnDec::PruneFrom(int bi, int bj)
{
    const int cols = (int)hlen_.size(), Pp = Pp_;
    int blen = 1, bprev = 0;
    if (bi >= 0 && packed_) {
        const int w = pk_[(size_t)bj * Pp + bi];
        blen = w & ((1 << kPackLenBits) - 1);
        bprev = (w >> kPackLenBits) - 1;
    } else if (bi >= 0) {
        blen = len_[(size_t)bj * Pp + bi];
        bprev = prev_[(size_t)bj * Pp + bi];
    }
    auto col = [&](int c) { const int k = hpos_ + c; return k >= cols ? k - cols : k; };
    int offs = cols - 1 - blen, phn = bprev;
    while (offs > 0) {
        const int k = col(offs);
        phn = hphn_[k];
        offs -= hlen_[k];
    }
    if (offs == 0) {                       // a phoneme ends exactly at the pruning horizon
        const int k0 = col(0);
        const int end = nframes_ - cols + 1, start = end - hlen_[k0];
        const float like = halpha_[k0] - prev_alpha_;
        prev_alpha_ = halpha_[k0];
        if (phn >= 0) labels_.push_back(Label{start, end, phn_[phn], like});
    }
}

void PhnDec::Done()
{
    const int cols = (int)hlen_.size();
    auto col = [&](int c) { const int k = hpos_ + c; return k >= cols ? k - cols : k; };
    int offs = cols - 1, end = nframes_;
    int phn = packed_ ? entry_prev_ : prev_[0];     // the winner that entered the loop last
    std::vector<Label> tail;
    while (offs > 0 && phn != -1) {
        const int k = col(offs);
        const int len = hlen_[k], start = end - len;
        const float a = halpha_[k];
        const int pphn = hphn_[k];
        offs -= len;
        const float like = offs > 0 ? a - halpha_[col(offs)] : a - prev_alpha_;
        tail.push_back(Label{start, end, phn_[phn], like});
        end = start;
        phn = pphn;
    }
    for (size_t i = tail.size(); i-- > 0;) labels_.push_back(tail[i]);
}

std::string FormatLabelLine(const Label &l)
{
    char buf[512];
    snprintf(buf, sizeof buf, "%d00000 %d00000 %s %f\n", l.start, l.end, l.phn.c_str(), l.score);
    return buf;
}

std::string FormatMlfLine(const Label &l)
{
    char a[32], b[32], buf[512];
    if (l.start == 0) snprintf(a, sizeof a, "0"); else snprintf(a, sizeof a, "%u00000", (unsigned)l.start);
    if (l.end == 0) snprintf(b, sizeof b, "0"); else snprintf(b, sizeof b, "%u00000", (unsigned)l.end);
    snprintf(buf, sizeof buf, "%s %s %s %f\n", a, b, l.phn.c_str(), l.score);
    return buf;
}

}  // namespace phnrec
