// frontend_dev.h -- parameter block of the GPU mel-bank front-end (frontend_kernels.hip)
#ifndef PHNREC_FRONTEND_DEV_H
#define PHNREC_FRONTEND_DEV_H

#include <hip/hip_runtime.h>

namespace phnrec {

struct FrontendParams {
    const unsigned char *bytes;   // raw utterances back to back (each at an even byte offset)
    const long long *sample_start; // [2*n_utts] first sample of each utterance, then its sample count
    const int *frame_off;         // [n_utts+1] first frame of each utterance
    float *mel;                   // [n_frames][nbanks]
    const float *hamming;         // [vector_size]
    const double *twiddle;        // [fft-1] (wr, wi) pairs: stage with half-size h starts at h-1
    const float *coeffs;          // [fft/2] mel filter weights per bin
    const int *run_begin;         // [2*nbanks_full] see melbank_kernel
    const int *run_end;
    int n_utts, n_frames, nbanks, fft;
    int wave_format;              // 1 lin16, 2 A-law
    int vector_size, vector_step;
    float dc_shift, scale, preem_coef;
    int z_mean_source;
    int raw_energies;             // 1: store the mel-bank energies themselves (the caller takes the logarithm: lcrc_wave_stage_energies)
    int ln_form;                  // LCRC_LN_*: how ln() is evaluated (lcrc_frontend_set_ln)
};

// the front-end's ln() (dspc.h:155-160: x > 0 ? logf(x) : 0) of n device values, in place; form = LCRC_LN_*
hipError_t frontend_ln_launch(float *x, size_t n, int form, hipStream_t stream);

hipError_t frontend_launch(const FrontendParams &p, hipStream_t stream);
hipError_t frontend_preload_code();
// host (pinned, mapped) -> device by a kernel instead of a copy command; both 16-byte aligned, `bytes` rounded up to 16
hipError_t pull_bytes_launch(const void *mapped_src, void *dst, size_t bytes, hipStream_t stream);
// Sentence mean normalisation.  means: device scratch [n_utts][nbanks].  block_off [n_utts + 1] (first
// 256-row block of each utterance, meannorm_blocks(rows) blocks each) and partial [n_blocks][nbanks] select the
// fixed-shape tree sum; block_off == NULL the reference's sequential sums.
int meannorm_blocks(int rows);
hipError_t meannorm_launch(float *mel, const int *frame_off, const int *block_off, int n_blocks, float *partial,
                           int n_utts, int n_rows, int nbanks, float *means, int max_utt_rows, hipStream_t stream);

}  // namespace phnrec
#endif
