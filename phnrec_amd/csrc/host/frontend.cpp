// frontend.cpp -- see frontend.h
#include "frontend.h"
#include "veclog.h"

#include "../meltables.h"

#include <immintrin.h>

#include <cmath>
#include <cstdlib>
#include <cstring>

namespace phnrec {

WaveFormat ParseWaveFormat(const std::string &s)
{
    if (s == "lin16") return WF_LIN16;
    if (s == "alaw") return WF_ALAW;
    return WF_UNKNOWN;
}

// ITU-T G.711 A-law expansion.  The reference keeps a 256-entry table of the expanded
// value divided by 8 (alaw.cpp:14-48) and multiplies by 8.0f (srec.cpp:769): the product
// is the G.711 value itself, which is what is computed here.
float ALawToLinear(unsigned char b)
{
    const unsigned a = b ^ 0x55u;
    int mant = (int)(a & 0x0Fu) << 4;
    const int seg = (int)(a & 0x70u) >> 4;
    if (seg == 0) mant += 8;
    else mant = (mant + 0x108) << (seg - 1);
    return (float)((a & 0x80u) ? mant : -mant);
}

void DecodeWaveform(const std::vector<unsigned char> &bytes, const WaveOptions &opt,
                    std::vector<float> &samples, int *n_samples)
{
    const int kMinLen = 200;                      // MB_VECTORSIZE (config.h:20)
    const int n = opt.format == WF_LIN16 ? (int)(bytes.size() / 2) : (int)bytes.size();
    samples.assign((size_t)(n > kMinLen ? n : kMinLen), 0.0f);
    if (opt.format == WF_LIN16) {
        for (int i = 0; i < n; i++) {
            short v;
            memcpy(&v, &bytes[2 * (size_t)i], 2);  // host byte order, as the reference's cast
            samples[i] = (float)v;
        }
    } else {
        for (int i = 0; i < n; i++) samples[i] = ALawToLinear(bytes[i]);
    }
    if (opt.dc_shift != 0.0f) for (int i = 0; i < n; i++) samples[i] = samples[i] + opt.dc_shift;
    if (opt.scale != 1.0f) for (int i = 0; i < n; i++) samples[i] = samples[i] * opt.scale;
    if (opt.noise_level != 0.0f)                 // sAddNoise dspc.h:100-105 (libc rand())
        for (int i = 0; i < n; i++)
            samples[i] += opt.noise_level * 2.0f * ((float)rand() / (float)RAND_MAX - 0.5f);
    *n_samples = n;
}

void MelBanks::Configure(int nbanks, int nbanks_full, int sample_freq, int vector_size, int step,
                         float preem_coef, bool z_mean_source, float lo_freq, float hi_freq)
{
    nbanks_ = nbanks; nbanks_full_ = nbanks_full; fs_ = sample_freq; vs_ = vector_size; step_ = step;
    preem_ = preem_coef; zmean_ = z_mean_source; lo_ = lo_freq; hi_ = hi_freq;
    init_ = false;
}

int MelBanks::NumFrames(int n) const { return n > vs_ ? (n - vs_) / step_ + 1 : 1; }

// Filter-bank design, window: shared with the GPU front-end (../meltables.cpp)
void MelBanks::Init()
{
    fft_ = FftSizeFor(vs_);
    BuildHamming(vs_, hamming_);
    if (nbanks_full_ == -1) nbanks_full_ = nbanks_;
    MelFilters f;
    BuildMelFilters(nbanks_full_, fft_, fs_, lo_, hi_, f);
    coeffs_ = f.coeffs;
    bank_of_ = f.bank_of;
    fftlo_ = f.fftlo;
    ffthi_ = f.ffthi;
    fft_buf_.assign(2 * (size_t)fft_ + 1, 0.0f);
    BuildTwiddles(fft_, twiddle_);        // the reference's per-frame recurrence, evaluated once
    en_.assign((size_t)nbanks_full_, 0.0f);
    init_ = true;
}

// In-place radix-2 decimation-in-time FFT on interleaved (re,im) pairs stored from index
// 1 (the classic "four1" arrangement the reference uses, dspc.cpp:24-78).  Parity needs
// its exact arithmetic: twiddles by the double-precision recurrence w += w*(wpr,wpi),
// butterfly products formed in double and rounded to float before the add/subtract.
static void Fft(float *d, unsigned nn, const double *tw)
{
    const unsigned n = nn << 1;
    for (unsigned i = 1, j = 1; i < n; i += 2) {          // bit reversal
        if (j > i) {
            float t = d[j]; d[j] = d[i]; d[i] = t;
            t = d[j + 1]; d[j + 1] = d[i + 1]; d[i + 1] = t;
        }
        unsigned m = n >> 1;
        while (m >= 2 && j > m) { j -= m; m >>= 1; }
        j += m;
    }
    // twiddle (stage of half-size h = span/2, butterfly k) = entry h - 1 + k of the table (meltables.cpp:
    // the recurrence w += w * (wpr, wpi) in double, exactly the values the reference forms per frame)
    for (unsigned span = 2; n > span; span <<= 1) {
        const unsigned stride = span << 1;
        const double *stage = tw + 2 * (size_t)(span / 2 - 1);
        for (unsigned m = 1; m < span; m += 2) {
            const double wr = stage[m - 1], wi = stage[m];
            for (unsigned i = m; i <= n; i += stride) {
                const unsigned j = i + span;
                const float tr = (float)(wr * d[j] - wi * d[j + 1]);
                const float ti = (float)(wr * d[j + 1] + wi * d[j]);
                d[j] = d[i] - tr;
                d[j + 1] = d[i + 1] - ti;
                d[i] += tr;
                d[i + 1] += ti;
            }
        }
    }
}

// MelBanks::ProcessFrame (melbanks.cpp:111-149) on a private copy of one frame.
void MelBanks::Frame(float *x, float *out)
{
    if (zmean_) {                                       // sSubtractAverage dspc.h:64-75
        float avg = 0.0f;
        for (int i = 0; i < vs_; i++) avg += x[i];
        avg /= (float)vs_;
        for (int i = 0; i < vs_; i++) x[i] -= avg;
    }
    if (preem_ != 0.0f) {                               // sPreemphasisBW dspc.h:77-84
        for (int n = vs_ - 1; n > 0; --n) x[n] -= preem_ * x[n - 1];
        x[0] *= (1.0f - preem_);
    }
    float *d = fft_buf_.data();
    d[0] = 0.0f;
    for (int i = 0; i < fft_; i++) {
        d[1 + 2 * i] = i < vs_ ? x[i] * hamming_[i] : 0.0f;
        d[2 + 2 * i] = 0.0f;
    }
    Fft(d, (unsigned)fft_, twiddle_.data());
    std::vector<float> &en = en_;
    std::fill(en.begin(), en.end(), 0.0f);
    for (int i = fftlo_; i <= ffthi_; i++) {            // _mbApply dspc.cpp:236-269
        const float re = d[1 + 2 * i], im = d[2 + 2 * i];
        const float p = re * re + im * im;              // cPower dspc.h:141-146
        const float v = coeffs_[i] * p;
        const int b = bank_of_[i];
        if (b > 0) en[b - 1] += v;
        if (b < nbanks_full_) en[b] += (p - v);
    }
    for (int b = 0; b < nbanks_; b++) out[b] = en[b] > 0.0f ? logf(en[b]) : 0.0f;   // sLn
}

// ---- eight frames in lockstep (AVX2) ------------------------------------------------------------------------------
// The front-end is the host's largest cost of a list run without -F (2 us per frame on one core: 16 of 17 CPU-seconds
// of a 10 000-file list).  Frames are independent, so eight of them go through Frame()'s operations side by side, one
// frame per vector lane: every lane performs exactly the scalar code's IEEE operations in the scalar code's order
// (separate multiplies and adds, the butterfly products in double and rounded to float, sequential sums, libm's
// scalar logf per value), so the features -- and the `-t par` dumps -- stay bit-identical to the reference's.
// Layout: value i of frame f at [i * 8 + f].
namespace {

bool UseAvx2()
{
    static const bool v = __builtin_cpu_supports("avx2") && !getenv("PHNREC_NO_AVX2");
    return v;
}

__attribute__((target("avx2")))
void Fft8(float *d, unsigned nn, const double *tw)
{
    const unsigned n = nn << 1;
    for (unsigned i = 1, j = 1; i < n; i += 2) {          // bit reversal: whole rows of eight
        if (j > i) {
            const __m256 a0 = _mm256_loadu_ps(d + 8 * j), a1 = _mm256_loadu_ps(d + 8 * (j + 1));
            _mm256_storeu_ps(d + 8 * j, _mm256_loadu_ps(d + 8 * i));
            _mm256_storeu_ps(d + 8 * (j + 1), _mm256_loadu_ps(d + 8 * (i + 1)));
            _mm256_storeu_ps(d + 8 * i, a0);
            _mm256_storeu_ps(d + 8 * (i + 1), a1);
        }
        unsigned m = n >> 1;
        while (m >= 2 && j > m) { j -= m; m >>= 1; }
        j += m;
    }
    for (unsigned span = 2; n > span; span <<= 1) {
        const unsigned stride = span << 1;
        const double *stage = tw + 2 * (size_t)(span / 2 - 1);
        // The first butterfly of every stage has the twiddle (1, 0) exactly (the recurrence starts there): its
        // (float)(1.0 * re - 0.0 * im) is `re` itself and (float)(1.0 * im + 0.0 * re) is `im` -- up to the sign of a
        // zero, which no later operation can turn into a different value (zeros only meet sums, products and the
        // power spectrum's squares).  A quarter of all butterflies: they skip the conversions and the products.
        for (unsigned i = 1; i <= n; i += stride) {
            const unsigned j = i + span;
            const __m256 tr = _mm256_loadu_ps(d + 8 * j), ti = _mm256_loadu_ps(d + 8 * (j + 1));
            const __m256 ar = _mm256_loadu_ps(d + 8 * i), ai = _mm256_loadu_ps(d + 8 * (i + 1));
            _mm256_storeu_ps(d + 8 * j, _mm256_sub_ps(ar, tr));
            _mm256_storeu_ps(d + 8 * (j + 1), _mm256_sub_ps(ai, ti));
            _mm256_storeu_ps(d + 8 * i, _mm256_add_ps(ar, tr));
            _mm256_storeu_ps(d + 8 * (i + 1), _mm256_add_ps(ai, ti));
        }
        for (unsigned m = 3; m < span; m += 2) {
            const __m256d wr = _mm256_set1_pd(stage[m - 1]), wi = _mm256_set1_pd(stage[m]);
            for (unsigned i = m; i <= n; i += stride) {
                const unsigned j = i + span;
                const __m256 xr = _mm256_loadu_ps(d + 8 * j), xi = _mm256_loadu_ps(d + 8 * (j + 1));
                const __m256d xr0 = _mm256_cvtps_pd(_mm256_castps256_ps128(xr)), xr1 = _mm256_cvtps_pd(_mm256_extractf128_ps(xr, 1));
                const __m256d xi0 = _mm256_cvtps_pd(_mm256_castps256_ps128(xi)), xi1 = _mm256_cvtps_pd(_mm256_extractf128_ps(xi, 1));
                // tr = (float)(wr * d[j] - wi * d[j+1]),  ti = (float)(wr * d[j+1] + wi * d[j]): products and sum in double
                const __m128 tr0 = _mm256_cvtpd_ps(_mm256_sub_pd(_mm256_mul_pd(wr, xr0), _mm256_mul_pd(wi, xi0)));
                const __m128 tr1 = _mm256_cvtpd_ps(_mm256_sub_pd(_mm256_mul_pd(wr, xr1), _mm256_mul_pd(wi, xi1)));
                const __m128 ti0 = _mm256_cvtpd_ps(_mm256_add_pd(_mm256_mul_pd(wr, xi0), _mm256_mul_pd(wi, xr0)));
                const __m128 ti1 = _mm256_cvtpd_ps(_mm256_add_pd(_mm256_mul_pd(wr, xi1), _mm256_mul_pd(wi, xr1)));
                const __m256 tr = _mm256_insertf128_ps(_mm256_castps128_ps256(tr0), tr1, 1);
                const __m256 ti = _mm256_insertf128_ps(_mm256_castps128_ps256(ti0), ti1, 1);
                const __m256 ar = _mm256_loadu_ps(d + 8 * i), ai = _mm256_loadu_ps(d + 8 * (i + 1));
                _mm256_storeu_ps(d + 8 * j, _mm256_sub_ps(ar, tr));
                _mm256_storeu_ps(d + 8 * (j + 1), _mm256_sub_ps(ai, ti));
                _mm256_storeu_ps(d + 8 * i, _mm256_add_ps(ar, tr));
                _mm256_storeu_ps(d + 8 * (i + 1), _mm256_add_ps(ai, ti));
            }
        }
    }
}

// Sixteen frames per step where the host has AVX-512 (the same operations once more, one frame per lane of a 512-bit
// register; PHNREC_NO_AVX512=1 keeps to eight).
bool UseAvx512()
{
    static const bool v = UseAvx2() && __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512dq") &&
                          !getenv("PHNREC_NO_AVX512");
    return v;
}

__attribute__((target("avx512f,avx512dq")))
void Fft16(float *d, unsigned nn, const double *tw)
{
    const unsigned n = nn << 1;
    for (unsigned i = 1, j = 1; i < n; i += 2) {          // bit reversal: whole rows of sixteen
        if (j > i) {
            const __m512 a0 = _mm512_loadu_ps(d + 16 * j), a1 = _mm512_loadu_ps(d + 16 * (j + 1));
            _mm512_storeu_ps(d + 16 * j, _mm512_loadu_ps(d + 16 * i));
            _mm512_storeu_ps(d + 16 * (j + 1), _mm512_loadu_ps(d + 16 * (i + 1)));
            _mm512_storeu_ps(d + 16 * i, a0);
            _mm512_storeu_ps(d + 16 * (i + 1), a1);
        }
        unsigned m = n >> 1;
        while (m >= 2 && j > m) { j -= m; m >>= 1; }
        j += m;
    }
    for (unsigned span = 2; n > span; span <<= 1) {
        const unsigned stride = span << 1;
        const double *stage = tw + 2 * (size_t)(span / 2 - 1);
        for (unsigned i = 1; i <= n; i += stride) {         // twiddle (1, 0): see Fft8
            const unsigned j = i + span;
            const __m512 tr = _mm512_loadu_ps(d + 16 * j), ti = _mm512_loadu_ps(d + 16 * (j + 1));
            const __m512 ar = _mm512_loadu_ps(d + 16 * i), ai = _mm512_loadu_ps(d + 16 * (i + 1));
            _mm512_storeu_ps(d + 16 * j, _mm512_sub_ps(ar, tr));
            _mm512_storeu_ps(d + 16 * (j + 1), _mm512_sub_ps(ai, ti));
            _mm512_storeu_ps(d + 16 * i, _mm512_add_ps(ar, tr));
            _mm512_storeu_ps(d + 16 * (i + 1), _mm512_add_ps(ai, ti));
        }
        for (unsigned m = 3; m < span; m += 2) {
            const __m512d wr = _mm512_set1_pd(stage[m - 1]), wi = _mm512_set1_pd(stage[m]);
            for (unsigned i = m; i <= n; i += stride) {
                const unsigned j = i + span;
                const __m512 xr = _mm512_loadu_ps(d + 16 * j), xi = _mm512_loadu_ps(d + 16 * (j + 1));
                const __m512d xr0 = _mm512_cvtps_pd(_mm512_castps512_ps256(xr)), xr1 = _mm512_cvtps_pd(_mm512_extractf32x8_ps(xr, 1));
                const __m512d xi0 = _mm512_cvtps_pd(_mm512_castps512_ps256(xi)), xi1 = _mm512_cvtps_pd(_mm512_extractf32x8_ps(xi, 1));
                // tr = (float)(wr * d[j] - wi * d[j+1]),  ti = (float)(wr * d[j+1] + wi * d[j]): products and sum in double
                const __m256 tr0 = _mm512_cvtpd_ps(_mm512_sub_pd(_mm512_mul_pd(wr, xr0), _mm512_mul_pd(wi, xi0)));
                const __m256 tr1 = _mm512_cvtpd_ps(_mm512_sub_pd(_mm512_mul_pd(wr, xr1), _mm512_mul_pd(wi, xi1)));
                const __m256 ti0 = _mm512_cvtpd_ps(_mm512_add_pd(_mm512_mul_pd(wr, xi0), _mm512_mul_pd(wi, xr0)));
                const __m256 ti1 = _mm512_cvtpd_ps(_mm512_add_pd(_mm512_mul_pd(wr, xi1), _mm512_mul_pd(wi, xr1)));
                const __m512 tr = _mm512_insertf32x8(_mm512_castps256_ps512(tr0), tr1, 1);
                const __m512 ti = _mm512_insertf32x8(_mm512_castps256_ps512(ti0), ti1, 1);
                const __m512 ar = _mm512_loadu_ps(d + 16 * i), ai = _mm512_loadu_ps(d + 16 * (i + 1));
                _mm512_storeu_ps(d + 16 * j, _mm512_sub_ps(ar, tr));
                _mm512_storeu_ps(d + 16 * (j + 1), _mm512_sub_ps(ai, ti));
                _mm512_storeu_ps(d + 16 * i, _mm512_add_ps(ar, tr));
                _mm512_storeu_ps(d + 16 * (i + 1), _mm512_add_ps(ai, ti));
            }
        }
    }
}

}  // namespace

// ... and for sixteen frames (AVX-512): layout [i * 16 + f]
__attribute__((target("avx512f,avx512dq")))
void MelBanks::Frame16(const float *s, float *out)
{
    float *x = x8_.data(), *d = d8_.data(), *en = en8_.data();
    for (int i = 0; i < vs_; i++)
        for (int f = 0; f < 16; f++) x[16 * i + f] = s[(size_t)f * step_ + i];
    if (zmean_) {                                       // sSubtractAverage dspc.h:64-75
        __m512 avg = _mm512_setzero_ps();
        for (int i = 0; i < vs_; i++) avg = _mm512_add_ps(avg, _mm512_loadu_ps(x + 16 * i));
        avg = _mm512_div_ps(avg, _mm512_set1_ps((float)vs_));
        for (int i = 0; i < vs_; i++) _mm512_storeu_ps(x + 16 * i, _mm512_sub_ps(_mm512_loadu_ps(x + 16 * i), avg));
    }
    if (preem_ != 0.0f) {                               // sPreemphasisBW dspc.h:77-84
        const __m512 pc = _mm512_set1_ps(preem_);
        for (int n = vs_ - 1; n > 0; --n)
            _mm512_storeu_ps(x + 16 * n, _mm512_sub_ps(_mm512_loadu_ps(x + 16 * n), _mm512_mul_ps(pc, _mm512_loadu_ps(x + 16 * (n - 1)))));
        _mm512_storeu_ps(x, _mm512_mul_ps(_mm512_loadu_ps(x), _mm512_set1_ps(1.0f - preem_)));
    }
    const __m512 zero = _mm512_setzero_ps();
    _mm512_storeu_ps(d, zero);
    for (int i = 0; i < fft_; i++) {
        _mm512_storeu_ps(d + 16 * (1 + 2 * i), i < vs_ ? _mm512_mul_ps(_mm512_loadu_ps(x + 16 * i), _mm512_set1_ps(hamming_[i])) : zero);
        _mm512_storeu_ps(d + 16 * (2 + 2 * i), zero);
    }
    Fft16(d, (unsigned)fft_, twiddle_.data());
    for (int b = 0; b < nbanks_full_; b++) _mm512_storeu_ps(en + 16 * b, zero);
    for (int i = fftlo_; i <= ffthi_; i++) {            // _mbApply dspc.cpp:236-269
        const __m512 re = _mm512_loadu_ps(d + 16 * (1 + 2 * i)), im = _mm512_loadu_ps(d + 16 * (2 + 2 * i));
        const __m512 p = _mm512_add_ps(_mm512_mul_ps(re, re), _mm512_mul_ps(im, im));     // cPower dspc.h:141-146
        const __m512 v = _mm512_mul_ps(_mm512_set1_ps(coeffs_[i]), p);
        const int b = bank_of_[i];
        if (b > 0) _mm512_storeu_ps(en + 16 * (b - 1), _mm512_add_ps(_mm512_loadu_ps(en + 16 * (b - 1)), v));
        if (b < nbanks_full_) _mm512_storeu_ps(en + 16 * b, _mm512_add_ps(_mm512_loadu_ps(en + 16 * b), _mm512_sub_ps(p, v)));
    }
    for (int f = 0; f < 16; f++)
        for (int b = 0; b < nbanks_; b++) out[(size_t)f * nbanks_ + b] = en[16 * b + f];
    LnInPlace(out, (size_t)16 * nbanks_);                // sLn: libm's logf, bit for bit, sixteen values at a time (veclog.cpp)
}

// MelBanks::ProcessFrame for the eight frames that start at s, s + step, ...: out[f * nbanks + b]
__attribute__((target("avx2")))
void MelBanks::Frame8(const float *s, float *out)
{
    float *x = x8_.data(), *d = d8_.data(), *en = en8_.data();
    for (int i = 0; i < vs_; i++)
        for (int f = 0; f < 8; f++) x[8 * i + f] = s[(size_t)f * step_ + i];
    if (zmean_) {                                       // sSubtractAverage dspc.h:64-75
        __m256 avg = _mm256_setzero_ps();
        for (int i = 0; i < vs_; i++) avg = _mm256_add_ps(avg, _mm256_loadu_ps(x + 8 * i));
        avg = _mm256_div_ps(avg, _mm256_set1_ps((float)vs_));
        for (int i = 0; i < vs_; i++) _mm256_storeu_ps(x + 8 * i, _mm256_sub_ps(_mm256_loadu_ps(x + 8 * i), avg));
    }
    if (preem_ != 0.0f) {                               // sPreemphasisBW dspc.h:77-84
        const __m256 pc = _mm256_set1_ps(preem_);
        for (int n = vs_ - 1; n > 0; --n)
            _mm256_storeu_ps(x + 8 * n, _mm256_sub_ps(_mm256_loadu_ps(x + 8 * n), _mm256_mul_ps(pc, _mm256_loadu_ps(x + 8 * (n - 1)))));
        _mm256_storeu_ps(x, _mm256_mul_ps(_mm256_loadu_ps(x), _mm256_set1_ps(1.0f - preem_)));
    }
    const __m256 zero = _mm256_setzero_ps();
    _mm256_storeu_ps(d, zero);
    for (int i = 0; i < fft_; i++) {
        _mm256_storeu_ps(d + 8 * (1 + 2 * i), i < vs_ ? _mm256_mul_ps(_mm256_loadu_ps(x + 8 * i), _mm256_set1_ps(hamming_[i])) : zero);
        _mm256_storeu_ps(d + 8 * (2 + 2 * i), zero);
    }
    Fft8(d, (unsigned)fft_, twiddle_.data());
    for (int b = 0; b < nbanks_full_; b++) _mm256_storeu_ps(en + 8 * b, zero);
    for (int i = fftlo_; i <= ffthi_; i++) {            // _mbApply dspc.cpp:236-269
        const __m256 re = _mm256_loadu_ps(d + 8 * (1 + 2 * i)), im = _mm256_loadu_ps(d + 8 * (2 + 2 * i));
        const __m256 p = _mm256_add_ps(_mm256_mul_ps(re, re), _mm256_mul_ps(im, im));     // cPower dspc.h:141-146
        const __m256 v = _mm256_mul_ps(_mm256_set1_ps(coeffs_[i]), p);
        const int b = bank_of_[i];
        if (b > 0) _mm256_storeu_ps(en + 8 * (b - 1), _mm256_add_ps(_mm256_loadu_ps(en + 8 * (b - 1)), v));
        if (b < nbanks_full_) _mm256_storeu_ps(en + 8 * b, _mm256_add_ps(_mm256_loadu_ps(en + 8 * b), _mm256_sub_ps(p, v)));
    }
    for (int f = 0; f < 8; f++)
        for (int b = 0; b < nbanks_; b++) out[(size_t)f * nbanks_ + b] = en[8 * b + f];
    LnInPlace(out, (size_t)8 * nbanks_);                 // sLn (veclog.cpp: libm's logf per value on hosts without AVX-512)
}

void MelBanks::Compute(std::vector<float> &samples, int n, std::vector<float> &out)
{
    if (!init_) Init();
    const int frames = NumFrames(n);
    // (the reference reads past its buffer when the signal is shorter than a 400-sample
    // frame; here the missing samples are zeros)
    if (samples.size() < (size_t)(frames - 1) * step_ + vs_) samples.resize((size_t)(frames - 1) * step_ + vs_, 0.0f);
    const float *s = samples.data();
    out.assign((size_t)frames * nbanks_, 0.0f);
    int t = 0;
    if (UseAvx2() && frames >= 8) {
        const size_t w = UseAvx512() && frames >= 16 ? 16 : 8;
        x8_.resize(w * (size_t)vs_);
        d8_.resize(w * (2 * (size_t)fft_ + 1));
        en8_.resize(w * (size_t)nbanks_full_);
        if (w == 16)
            for (; t + 16 <= frames; t += 16) Frame16(s + (size_t)t * step_, &out[(size_t)t * nbanks_]);
        for (; t + 8 <= frames; t += 8) Frame8(s + (size_t)t * step_, &out[(size_t)t * nbanks_]);
    }
    std::vector<float> frame(vs_);
    for (; t < frames; t++) {
        // frame t covers samples [t*step, t*step + vs) (the streaming copy/shift of
        // MelBanks::GetFeatures, melbanks.cpp:151-204, reduces to this)
        memcpy(frame.data(), s + (size_t)t * step_, sizeof(float) * vs_);
        Frame(frame.data(), &out[(size_t)t * nbanks_]);
    }
}

void SentenceMeanNorm(float *mel, int rows, int cols)
{
    for (int c = 0; c < cols; c++) {
        float sum = 0.0f;
        for (int r = 0; r < rows; r++) sum += mel[(size_t)r * cols + c];
        const float mean = sum * (1.0f / (float)rows);
        for (int r = 0; r < rows; r++) mel[(size_t)r * cols + c] += -mean;
    }
}

// offlinenorm/sent_max_norm and sent_chmax_norm (srec.cpp:1547-1587), after the mean normalisation: every column's
// maximum over the utterance (from -9999.9, strict >) is subtracted from the column.  With sent_max_norm the reference
// means the maximum over all columns, but its loop overwrites the whole row of maxima with the running value in EVERY
// iteration (`max.set(global_max)` inside the loop, :1573-1580): what is subtracted everywhere is column 0's maximum.
// Restated as it is.
void SentenceMaxNorm(float *mel, int rows, int cols, bool global)
{
    std::vector<float> mx((size_t)cols, -9999.9f);
    for (int c = 0; c < cols; c++)
        for (int r = 0; r < rows; r++) {
            const float v = mel[(size_t)r * cols + c];
            if (v > mx[c]) mx[c] = v;
        }
    if (global) {
        float g = -9999.9f;
        for (int c = 0; c < cols; c++) {
            if (mx[c] > g) g = mx[c];
            for (int k = 0; k < cols; k++) mx[k] = g;
        }
    }
    for (int c = 0; c < cols; c++)
        for (int r = 0; r < rows; r++) mel[(size_t)r * cols + c] = mel[(size_t)r * cols + c] - mx[c];
}

}  // namespace phnrec
