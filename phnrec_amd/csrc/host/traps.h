// traps.h -- host-side mirror of the reference's `class Traps` (traps.h:21-76) over the
// C ABI of libphnrec_lcrc.so.  Same method names, argument meaning and call order as the
// reference (setters, then Init(dir), then Reset()/CalcFeaturesBunched()), so code written
// against the reference class reads the same; the work happens in the fused HIP kernel.
// Differences, all deliberate: Init() reports failure by return value + LastError()
// instead of exit(1); CalcUtterance()/CalcBatch() expose the whole-utterance forms the
// GPU wants (ProcessOffline's prime/main/flush collapses into one launch).
#ifndef PHNREC_HOST_TRAPS_H
#define PHNREC_HOST_TRAPS_H

#include <string>

#include "../../../include/lcrc_experimental.h"      // lcrc.h (the Traps seam) + lcrc_pipeline.h (lists) + SetHiddenSplit / SetWaitMode

namespace phnrec {

class Traps {
public:
    Traps() = default;
    ~Traps() { if (ctx_) lcrc_destroy(ctx_); }
    Traps(const Traps &) = delete;
    Traps &operator=(const Traps &) = delete;

    // -- the reference's interface (traps.h:59-75) --
    bool SetSystem(const char *sys) { system_ = sys; return system_ == "LCRC" || system_ == "3BT" || system_ == "1BT" || system_ == "1BT_DCT"; }
    void SetTrapLen(int v) { trap_len_ = v; }
    void SetHamming(bool v) { hamming_ = v; }
    void SetNBanks(int v) { nbanks_ = v; }
    void SetAddC0(bool v) { add_c0_ = v; }
    void SetBunchSize(int v) { bunch_ = v; }        // grouping only; never changes values
    bool Init(const char *dir)
    {
        if (ctx_) { lcrc_destroy(ctx_); ctx_ = nullptr; }
        const int rc = lcrc_create_system(&ctx_, dir, system_.c_str(), nbanks_, trap_len_, add_c0_ ? 1 : 0,
                                          hamming_ ? 1 : 0, device_);
        if (rc != LCRC_OK) { err_ = lcrc_last_error(nullptr); ctx_ = nullptr; return false; }
        lcrc_set_hidden_split(ctx_, hidden_split_);
        return true;
    }
    // a second context over `src`'s model on the same GPU (lcrc_clone): shares its weights on the device
    bool InitClone(const Traps &src)
    {
        if (ctx_) { lcrc_destroy(ctx_); ctx_ = nullptr; }
        if (lcrc_clone(&ctx_, src.ctx_) != LCRC_OK) { err_ = lcrc_last_error(nullptr); ctx_ = nullptr; return false; }
        lcrc_set_hidden_split(ctx_, hidden_split_);
        return true;
    }
    void Reset() { lcrc_reset(ctx_); }
    void CalcFeaturesBunched(float *band_energies, float *features, int n = 1, bool neededFea = true)
    {
        if (lcrc_push(ctx_, band_energies, n, features, neededFea ? 1 : 0) != LCRC_OK) err_ = lcrc_last_error(ctx_);
    }
    void CalcFeatures(float *band_energies, float *features, int n = 1, bool neededFea = true)
    {
        CalcFeaturesBunched(band_energies, features, n, neededFea);
    }
    int GetNumOuts() const { return lcrc_num_outputs(ctx_); }
    int GetTrapShift() const { return (trap_len_ - 1) / 2; }
    int GetDelay() const { return lcrc_delay(ctx_); }

    // -- additions --
    void SetDevice(int d) { device_ = d; }
    // lcrc_set_hidden_split: 0 = small launches spread a tile's hidden units over several workgroups (default),
    // 1 = fused kernel only (a frame's output never depends on what else shares its launch)
    void SetHiddenSplit(int v) { hidden_split_ = v; if (ctx_) lcrc_set_hidden_split(ctx_, v); }
    // 0: the calling thread spins while the device works; n: it sleeps, looking every n microseconds (lcrc_set_wait_mode)
    void SetWaitMode(int poll_us) { if (ctx_) lcrc_set_wait_mode(ctx_, poll_us); }
    // fn(arg) on the calling thread as soon as a call's posterior kernels are done (lcrc_set_kernel_done_callback)
    void SetKernelDoneCallback(lcrc_kernel_done_fn fn, void *arg) { if (ctx_) lcrc_set_kernel_done_callback(ctx_, fn, arg); }
    // lcrc_set_arithmetic: false when the model has no split-f16 form (LastError() says why)
    bool SetArithmetic(int a)
    {
        if (lcrc_set_arithmetic(ctx_, a) == LCRC_OK) return true;
        err_ = lcrc_last_error(ctx_);
        return false;
    }
    bool CalcUtterance(const float *mel, int n, float *post)
    {
        if (lcrc_posteriors(ctx_, mel, n, post) == LCRC_OK) return true;
        err_ = lcrc_last_error(ctx_);
        return false;
    }
    // whether CalcRows exists for this context (the fused LCRC kernels; not the other systems / general geometries)
    bool HasRowRanges() const { return ctx_ && system_ == "LCRC" && std::string(lcrc_kernel_name(ctx_)) != "lcrc_general"; }
    // a row range of a strip (lcrc_posteriors_rows): chunks of a file longer than one launch, cut with their halos
    bool CalcRows(const float *strip, int n_rows, int row_first, int row_count, float *post)
    {
        if (lcrc_posteriors_rows(ctx_, strip, n_rows, row_first, row_count, post) == LCRC_OK) return true;
        err_ = lcrc_last_error(ctx_);
        return false;
    }
    bool CalcBatch(const float *mel, const int *off, int n_utts, float *post)
    {
        if (lcrc_posteriors_batch(ctx_, mel, off, n_utts, post) == LCRC_OK) return true;
        err_ = lcrc_last_error(ctx_);
        return false;
    }
    // zero-copy staging (lcrc_stage_buffers / lcrc_stage_run)
    bool StageBuffers(int rows, float **mel, float **post)
    {
        if (lcrc_stage_buffers(ctx_, rows, mel, post) == LCRC_OK) return true;
        err_ = lcrc_last_error(ctx_);
        return false;
    }
    bool StageRun(const int *off, int n_utts)
    {
        if (lcrc_stage_run(ctx_, off, n_utts) == LCRC_OK) return true;
        err_ = lcrc_last_error(ctx_);
        return false;
    }
    // waveform entry: mel-bank front-end on the GPU (lcrc_frontend_configure / lcrc_wave_to_posteriors)
    bool ConfigureFrontend(const lcrc_frontend &fe)
    {
        if (lcrc_frontend_configure(ctx_, &fe) == LCRC_OK) return true;
        err_ = lcrc_last_error(ctx_);
        return false;
    }
    // ln() of the GPU front-end: the host libm's own sequence (LCRC_LN_GLIBC_FMA / LCRC_LN_GLIBC) or log() in double (lcrc_frontend_set_ln)
    void SetFrontendLn(int form) { if (ctx_) lcrc_frontend_set_ln(ctx_, form); }
    int FrontendFrames(long long n_bytes) const { return lcrc_frontend_frames(ctx_, n_bytes); }
    bool WaveToPosteriors(const unsigned char *bytes, const long long *byte_off, int n_utts, float *post, int *frame_off)
    {
        if (lcrc_wave_to_posteriors(ctx_, bytes, byte_off, n_utts, post, frame_off) == LCRC_OK) return true;
        err_ = lcrc_last_error(ctx_);
        return false;
    }
    // posterior writer path (lcrc_output_configure): softening and byte order on the device
    bool ConfigureOutput(const lcrc_softening *stages, int n_stages, bool big_endian)
    {
        if (lcrc_output_configure(ctx_, stages, n_stages, big_endian ? 1 : 0) == LCRC_OK) return true;
        err_ = lcrc_last_error(ctx_);
        return false;
    }
    // decoder on the device (lcrc_decoder_configure / lcrc_last_labels)
    bool ConfigureDecoder(int n_phonemes, int states, int time_pruning, float wpenalty, bool posterior_readback)
    {
        if (lcrc_decoder_configure(ctx_, n_phonemes, states, time_pruning, wpenalty) == LCRC_OK &&
            lcrc_set_posterior_readback(ctx_, posterior_readback ? 1 : 0) == LCRC_OK) return true;
        err_ = lcrc_last_error(ctx_);
        return false;
    }
    bool LastLabels(const lcrc_label **labels, const int **first, const int **count, int *n_utts)
    {
        return lcrc_last_labels(ctx_, labels, first, count, n_utts) == LCRC_OK;
    }
    // the decoder of a staged call beside the next call's kernels; that call's labels: PrevLabels after the next call has
    // returned, LastLabels after the last one (lcrc_set_decoder_overlap / lcrc_prev_labels)
    bool SetDecoderOverlap(bool on)
    {
        if (lcrc_set_decoder_overlap(ctx_, on ? 1 : 0) == LCRC_OK) return true;
        err_ = lcrc_last_error(ctx_);
        return false;
    }
    // posterior kernels of this device's contexts one after the other, in queueing order (lcrc_set_launch_order)
    void SetLaunchOrder(bool on) { if (ctx_) lcrc_set_launch_order(ctx_, on ? 1 : 0); }
    bool PrevLabels(const lcrc_label **labels, const int **first, const int **count, int *n_utts)
    {
        return lcrc_prev_labels(ctx_, labels, first, count, n_utts) == LCRC_OK;
    }
    // buffers for launches of up to this size, allocated ahead of the first one (lcrc_reserve)
    bool Reserve(int max_rows, int max_utts, long long max_wave_bytes)
    {
        if (lcrc_reserve(ctx_, max_rows, max_utts, max_wave_bytes) == LCRC_OK) return true;
        err_ = lcrc_last_error(ctx_);
        return false;
    }
    // zero-copy waveform staging (lcrc_wave_stage_buffer / lcrc_wave_stage_run)
    bool WaveStageBuffer(long long capacity, unsigned char **bytes)
    {
        if (lcrc_wave_stage_buffer(ctx_, capacity, bytes) == LCRC_OK) return true;
        err_ = lcrc_last_error(ctx_);
        return false;
    }
    bool WaveStageRun(const long long *start, const long long *n_bytes, int n_utts, float *post, int *frame_off)
    {
        if (lcrc_wave_stage_run(ctx_, start, n_bytes, n_utts, post, frame_off) == LCRC_OK) return true;
        err_ = lcrc_last_error(ctx_);
        return false;
    }
    // the front-end up to the mel-bank energies (lcrc_wave_stage_energies): they come back in the pinned feature buffer
    bool WaveStageEnergies(const long long *start, const long long *n_bytes, int n_utts, float **energies, int *frame_off)
    {
        if (lcrc_wave_stage_energies(ctx_, start, n_bytes, n_utts, energies, frame_off) == LCRC_OK) return true;
        err_ = lcrc_last_error(ctx_);
        return false;
    }
    const float *StagedPosteriors() { const float *p = nullptr; lcrc_staged_posteriors(ctx_, &p); return p; }
    float LastKernelMs() { float ms = 0; lcrc_last_kernel_ms(ctx_, &ms); return ms; }
    const std::string &LastError() const { return err_; }
    bool Ready() const { return ctx_ != nullptr; }

private:
    lcrc_ctx *ctx_ = nullptr;
    std::string system_ = "LCRC", err_;
    int trap_len_ = 31, nbanks_ = 15, bunch_ = 1, device_ = 0, hidden_split_ = 0;
    bool hamming_ = false, add_c0_ = true;
};

}  // namespace phnrec
#endif
