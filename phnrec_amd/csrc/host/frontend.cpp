// frontend.cpp -- see frontend.h
#include "frontend.h"

#include "../meltables.h"

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

void MelBanks::Compute(std::vector<float> &samples, int n, std::vector<float> &out)
{
    if (!init_) Init();
    const int frames = NumFrames(n);
    // (the reference reads past its buffer when the signal is shorter than a 400-sample
    // frame; here the missing samples are zeros)
    if (samples.size() < (size_t)(frames - 1) * step_ + vs_) samples.resize((size_t)(frames - 1) * step_ + vs_, 0.0f);
    const float *s = samples.data();
    out.assign((size_t)frames * nbanks_, 0.0f);
    std::vector<float> frame(vs_);
    for (int t = 0; t < frames; t++) {
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

}  // namespace phnrec
