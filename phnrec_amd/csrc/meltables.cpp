// meltables.cpp -- see meltables.h
#include "meltables.h"

#include <cmath>

namespace phnrec {

int FftSizeFor(int vs)
{
    int n = 1;
    while (n < vs) n *= 2;
    return n;
}

void BuildHamming(int vs, std::vector<float> &w)
{
    w.resize(vs);
    for (int i = 0; i < vs; i++)
        w[i] = 1.0f * (0.54f - 0.46f * cosf(2.0f * (float)M_PI * i / (vs - 1)));
}

static inline float MelOf(float f) { return 1127.0f * logf(1.0f + f / 700.0f); }   // dspc.h:174-177

// `count` triangular filters equally spaced on the mel scale between fmin and fmax; every FFT bin
// i in [fftlo, ffthi] belongs to the falling edge of filter bank_of[i]-1 with weight coeffs[i] and to
// the rising edge of filter bank_of[i] with 1-coeffs[i].  Centres are accumulated by repeated f32
// addition, bins are assigned by a monotone scan -- both as in the reference.
void BuildMelFilters(int count, int fft, int fs, float fmin, float fmax, MelFilters &f)
{
    const int half = fft / 2;
    if (fmin < 0.0f) fmin = 0.0f;
    if (fmax > (float)fs / 2.0f) fmax = (float)fs / 2.0f;
    f.count = count; f.fft = fft;
    f.coeffs.assign(half, 0.0f);
    f.bank_of.assign(half, -1);
    const float bf = (float)fs / (float)fft;
    const float mlo = MelOf(fmin), mhi = MelOf(fmax);
    f.fftlo = (int)(fmin / bf + 1.5f);
    f.ffthi = (int)(fmax / bf - 0.5f);
    if (f.fftlo < 1) f.fftlo = 1;
    if (f.ffthi >= half) f.ffthi = half - 1;
    const float delta = (mhi - mlo) / (count + 1);
    std::vector<float> centre(count + 1);
    float m = mlo;
    for (int i = 0; i <= count; i++) { m = m + delta; centre[i] = m; }
    int ch = 0;
    for (int i = f.fftlo; i <= f.ffthi; i++) {
        const float mf = MelOf((float)i * bf);
        // (index test first: `centre` has count + 1 entries; a bin beyond the last centre -- only reachable
        //  through rounding at the very top of the band -- stays with the last filter's falling edge)
        while (ch <= count && mf > centre[ch]) ++ch;
        if (ch > count) ch = count;
        f.bank_of[i] = (short)ch;
    }
    for (int i = f.fftlo; i <= f.ffthi; i++) {
        const int c = f.bank_of[i];
        const float mf = MelOf((float)i * bf);
        f.coeffs[i] = c == 0 ? (centre[0] - mf) / (centre[0] - mlo) : (centre[c] - mf) / (centre[c] - centre[c - 1]);
    }
}

void BuildTwiddles(int fft, std::vector<double> &tw)
{
    tw.assign(2 * (size_t)(fft - 1), 0.0);
    for (int h = 1; h < fft; h <<= 1) {
        const unsigned span = 2u * h;                           // "mmax" in floats of the interleaved array
        const double theta = -(6.28318530717959 / span);        // forward transform
        const double s = sin(0.5 * theta);
        const double wpr = -2.0 * s * s, wpi = sin(theta);
        double wr = 1.0, wi = 0.0;
        for (int k = 0; k < h; k++) {
            tw[2 * (size_t)(h - 1 + k)] = wr;
            tw[2 * (size_t)(h - 1 + k) + 1] = wi;
            const double t = wr;
            wr = t * wpr - wi * wpi + t;
            wi = wi * wpr + t * wpi + wi;
        }
    }
}

}  // namespace phnrec
