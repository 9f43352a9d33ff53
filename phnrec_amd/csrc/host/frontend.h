// frontend.h -- host front-end of the drop-in CLI: raw waveform -> log mel-bank energies.
//
// Not on the GPU hot path (0.3 % of the reference's run time): it stays on
// the host and reproduces the reference's arithmetic so that `-t par` dumps and the
// label goldens match: waveform decode srec.cpp:709-791 + alaw.cpp, framing srec.cpp:945 /
// melbanks.cpp:151-204, Hamming dspc.h:162-167, FFT dspc.cpp:24-78, mel filters
// dspc.cpp:80-269, ln dspc.h:155-160.
#ifndef PHNREC_HOST_FRONTEND_H
#define PHNREC_HOST_FRONTEND_H

#include <string>
#include <vector>

namespace phnrec {

enum WaveFormat { WF_UNKNOWN = 0, WF_LIN16, WF_ALAW };
WaveFormat ParseWaveFormat(const std::string &s);        // "lin16" | "alaw"

// A-law byte -> linear sample as the reference's table gives it (8 * ALawTableD5[b]).
float ALawToLinear(unsigned char b);

struct WaveOptions {
    WaveFormat format = WF_LIN16;
    float scale = 1.0f, dc_shift = 0.0f, noise_level = 0.0f;
};

// Raw bytes (no header parsing: a WAV header is consumed as samples, like the reference)
// -> float samples.  The buffer holds at least 200 samples, the first 200 zero-initialised
// before the copy (MB_VECTORSIZE, srec.cpp:731), so short signals still give one frame.
void DecodeWaveform(const std::vector<unsigned char> &bytes, const WaveOptions &opt,
                    std::vector<float> &samples, int *n_samples);

class MelBanks {
public:
    void Configure(int nbanks, int nbanks_full, int sample_freq, int vector_size, int step,
                   float preem_coef, bool z_mean_source, float lo_freq, float hi_freq);
    int NumBanks() const { return nbanks_; }
    int VectorSize() const { return vs_; }
    int Step() const { return step_; }
    // nFrames = len > vs ? (len - vs)/step + 1 : 1   (srec.cpp:945)
    int NumFrames(int n_samples) const;
    // out [NumFrames][nbanks].  `samples` is zero-extended to one full frame if shorter.
    void Compute(std::vector<float> &samples, int n_samples, std::vector<float> &out);

private:
    void Init();
    void Frame(float *frame, float *out);
    void Frame8(const float *first, float *out);      // eight consecutive frames in lockstep (AVX2), same arithmetic per frame
    void Frame16(const float *first, float *out);     // sixteen (AVX-512)
    int nbanks_ = 15, nbanks_full_ = -1, fs_ = 8000, vs_ = 200, step_ = 80, fft_ = 256;
    float preem_ = 0.0f, lo_ = 64.0f, hi_ = 4000.0f;
    bool zmean_ = false, init_ = false;
    std::vector<float> hamming_, coeffs_, fft_buf_, en_;
    std::vector<float> x8_, d8_, en8_;               // scratch of Frame8 / Frame16: [index][8 or 16 frames]
    std::vector<double> twiddle_;
    std::vector<short> bank_of_;
    int fftlo_ = 0, ffthi_ = 0;
};

// offlinenorm/sent_mean_norm (srec.cpp:1500-1511): column sums are sequential f32,
// mean = sum * (1.0f / rows), x += -mean.
void SentenceMeanNorm(float *mel, int rows, int cols);
// offlinenorm/sent_chmax_norm (global = false) and sent_max_norm (global = true) as the reference computes them
void SentenceMaxNorm(float *mel, int rows, int cols, bool global);

}  // namespace phnrec
#endif
