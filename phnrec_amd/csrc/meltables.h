// meltables.h -- tables of the mel-bank front-end, built on the host with the reference's own
// expressions (so host and GPU front-ends share bit-identical constants):
//   Hamming window   dspc.h:162-167
//   mel filter bank  dspc.cpp:80-197 (_mbInit)
//   FFT twiddles     dspc.cpp:54-74  (double-precision recurrence of "four1")
#ifndef PHNREC_MELTABLES_H
#define PHNREC_MELTABLES_H

#include <vector>

namespace phnrec {

struct MelFilters {
    int count = 0, fft = 0, fftlo = 0, ffthi = 0;
    std::vector<float> coeffs;     // [fft/2] weight of bin i on the falling edge of filter bank_of[i]-1
    std::vector<short> bank_of;    // [fft/2] -1 outside [fftlo, ffthi]
};

int FftSizeFor(int vector_size);                                  // next power of two >= vector_size
void BuildHamming(int vector_size, std::vector<float> &w);
void BuildMelFilters(int count, int fft, int sample_freq, float fmin, float fmax, MelFilters &f);
// (wr, wi) for every (stage, k): the stage whose butterflies span h complex elements starts at h-1
void BuildTwiddles(int fft, std::vector<double> &tw);

}  // namespace phnrec
#endif
