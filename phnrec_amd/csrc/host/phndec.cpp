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

}  // namespace

void PhnDec::ProcessFrame(const float *f)
{
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
    int blen = 1, bprev = 0;
    if (bi >= 0) {
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
    int phn = prev_[0];                    // the winner that entered the loop last
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
