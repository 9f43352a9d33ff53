// lcrc_api_decoder.cpp -- the decoder on the device behind the C ABI ("next" row f3; include/lcrc.h lcrc_decoder_configure ...):
// label buffers, the launch of phndec_kernel behind the posterior kernels, and the overlap mode in which a staged call's
// decoder runs beside the context's next call.  The kernel itself: phndec_kernels.hip.
#include "lcrc_ctx.h"

namespace lcrc_impl {

// Label buffers: pinned host memory mapped into the device -- the decoder kernel stores its labels and counts where the
// host reads them (16 B per label, one per ~8 frames: posted writes over PCIe), so a launch queues NO copy command for
// them.  As copy commands they stood in the device's copy queue, which every context shares in order, behind their 2 ms
// decoder kernel -- and the next launch's upload of its files behind them (profiles/r05_ab_runs.txt 3).
// d_labels / d_count are the device's view of h_labels / h_count.
int ensure_labels(lcrc_ctx *c, size_t n_rows, size_t n_utts)
{
    if (n_rows > c->cap_label_rows) {
        const size_t cap = n_rows + n_rows / 4 + 64;
        if (c->h_labels) (void)pinned_free(c->h_labels);
        c->d_labels = c->h_labels = nullptr;
        c->cap_label_rows = 0;
        HIP_TRY(c, pinned_alloc((void **)&c->h_labels, cap * sizeof(lcrc_label), true));
        HIP_TRY(c, hipHostGetDevicePointer((void **)&c->d_labels, c->h_labels, 0));
        c->cap_label_rows = cap;
    }
    if (n_utts > c->cap_label_utts) {
        const size_t cap = n_utts + n_utts / 4 + 64;
        if (c->h_count) (void)pinned_free(c->h_count);
        c->d_count = c->h_count = nullptr;
        c->cap_label_utts = 0;
        HIP_TRY(c, pinned_alloc((void **)&c->h_count, cap * sizeof(int), true));
        HIP_TRY(c, hipHostGetDevicePointer((void **)&c->d_count, c->h_count, 0));
        c->cap_label_utts = cap;
    }
    return LCRC_OK;
}

// lcrc_set_decoder_overlap applies to the staged entry points of a context that decodes without reading posteriors back
bool overlap_on(const lcrc_ctx *c) { return c->dec_overlap && c->dec_P > 0 && !c->readback; }

void swap_decoder_sets(lcrc_ctx *c)
{
    std::swap(c->d_post, c->alt.d_post);
    std::swap(c->d_post_cap, c->alt.d_post_cap);
    std::swap(c->d_labels, c->alt.d_labels);
    std::swap(c->h_labels, c->alt.h_labels);
    std::swap(c->d_count, c->alt.d_count);
    std::swap(c->h_count, c->alt.h_count);
    std::swap(c->cap_label_rows, c->alt.cap_label_rows);
    std::swap(c->cap_label_utts, c->alt.cap_label_utts);
    c->label_first.swap(c->alt.label_first);
    std::swap(c->label_utts, c->alt.label_utts);
    std::swap(c->d_dec_off, c->alt.d_dec_off);
    std::swap(c->cap_dec_off, c->alt.cap_dec_off);
    std::swap(c->ev_dec_done, c->alt.ev_dec_done);
    std::swap(c->dec_pending, c->alt.dec_pending);
}

// d_post for as many rows as the rest of the frame staging holds (the two sets' buffers grow one call apart)
int ensure_post_rows(lcrc_ctx *c)
{
    if (c->d_post_cap >= c->cap_rows) return LCRC_OK;
    if (c->d_post) (void)hipFree(c->d_post);
    c->d_post = nullptr;
    c->d_post_cap = 0;
    if (dev_alloc((void **)&c->d_post, c->cap_rows * (size_t)c->nets[2].n_out * sizeof(float)) != hipSuccess) {
        (void)hipGetLastError();
        return fail(c, LCRC_E_NOMEM, "cannot allocate the second posterior buffer (lcrc_set_decoder_overlap)");
    }
    c->d_post_cap = c->cap_rows;
    return LCRC_OK;
}

// A decoder that an overlapped staged call left running is ordered only against LATER overlapped calls (the event waits
// below).  Every other call on the context -- the synchronous entry points, a staged call after lcrc_set_posterior_readback(1)
// or lcrc_decoder_configure(0) -- writes d_post on the context's stream and may start a decoder of its own that stores into
// the same label buffers: such a call first waits here for both sets' decoders (lcrc.h: mixing entry points is allowed).
int settle_pending_decoders(lcrc_ctx *c)
{
    if (!c->dec_pending && !c->alt.dec_pending) return LCRC_OK;
    if (c->dec_stream) HIP_TRY(c, hipStreamSynchronize(c->dec_stream));
    c->dec_pending = c->alt.dec_pending = false;
    return LCRC_OK;
}

// Start of a staged call under lcrc_set_decoder_overlap: the call works on the set that the call BEFORE the last one used,
// while the last call's decoder may still be reading and writing the other.
int begin_overlapped_call(lcrc_ctx *c)
{
    c->overlapped_call = overlap_on(c);
    if (!c->overlapped_call) return settle_pending_decoders(c);
    swap_decoder_sets(c);
    // what this call's kernels overwrite was read by the decoder two calls ago: behind it on the device, whether or not
    // the caller has fetched those labels
    if (c->dec_pending) HIP_TRY(c, hipStreamWaitEvent(c->stream, c->ev_dec_done, 0));
    c->dec_pending = false;
    c->label_utts = 0;
    return c->cap_rows > 0 ? ensure_post_rows(c) : LCRC_OK;
}

// Decoder behind the posterior kernel (it stores labels and counts in the pinned buffers itself).  On the launch stream `s`;
// `staged` calls of a context under lcrc_set_decoder_overlap queue it on the decoder stream instead, behind an event on `s`,
// and return without waiting for it (lcrc_last_labels / lcrc_prev_labels wait).
// h_first: host copy of the utterance offsets (first label slot of each utterance).
int decode_after(lcrc_ctx *c, const int *d_off, const int *h_first, int n_utts, int n_rows, const float *d_post,
                 hipStream_t s, bool staged)
{
    c->label_utts = 0;
    if (c->dec_P <= 0 || n_rows <= 0) return LCRC_OK;
    if (c->out_be) return fail(c, LCRC_E_ARG, "the decoder needs posteriors in host byte order (lcrc_output_configure big_endian=0)");
    const bool overlap = staged && overlap_on(c);
    if (!overlap) { const int rc = settle_pending_decoders(c); if (rc) return rc; }      // (ensure_labels may free what a decoder still writes)
    { const int rc = ensure_labels(c, (size_t)n_rows, (size_t)n_utts); if (rc) return rc; }
    hipStream_t ds = s;
    if (overlap) {
        if ((size_t)n_utts + 1 > c->cap_dec_off) {
            const size_t cap = (size_t)n_utts + n_utts / 4 + 64;
            if (c->d_dec_off) (void)hipFree(c->d_dec_off);
            c->d_dec_off = nullptr;
            c->cap_dec_off = 0;
            HIP_TRY(c, hipMalloc((void **)&c->d_dec_off, cap * sizeof(int)));
            c->cap_dec_off = cap;
        }
        HIP_TRY(c, hipMemcpyAsync(c->d_dec_off, d_off, (size_t)(n_utts + 1) * sizeof(int), hipMemcpyDeviceToDevice, s));
        HIP_TRY(c, hipEventRecord(c->ev_post, s));
        HIP_TRY(c, hipStreamWaitEvent(c->dec_stream, c->ev_post, 0));
        d_off = c->d_dec_off;
        ds = c->dec_stream;
    }
    PhnDecParams p;
    memset(&p, 0, sizeof p);
    p.logpost = d_post; p.off = d_off; p.n_utts = n_utts; p.cols = c->nets[2].n_out;
    p.P = c->dec_P; p.S = c->dec_S; p.prune = c->dec_prune; p.wpen = c->dec_wpen;
    p.labels = c->d_labels; p.count = c->d_count;
    HIP_TRY(c, phndec_launch(p, ds));          // (labels and counts: stored by the kernel straight into the pinned buffers)
    if (overlap) {
        HIP_TRY(c, hipEventRecord(c->ev_dec_done, ds));
        c->dec_pending = true;
    }
    c->label_first.assign(h_first, h_first + n_utts);
    c->label_utts = n_utts;
    return LCRC_OK;
}

}  // namespace lcrc_impl

using namespace lcrc_impl;

extern "C" {

int lcrc_decoder_configure(lcrc_ctx *c, int n_phonemes, int states_per_phn, int time_pruning, float wpenalty)
{
    if (!c) return LCRC_E_ARG;
    if (n_phonemes == 0) { c->dec_P = 0; c->label_utts = 0; return LCRC_OK; }
    if (n_phonemes < 0 || n_phonemes > 64 || states_per_phn < 1 || states_per_phn > 4 || time_pruning < 1 ||
        time_pruning > 63 || n_phonemes * states_per_phn > c->nets[2].n_out)
        return fail(c, LCRC_E_UNSUPPORTED, "lcrc_decoder_configure: needs <= 64 phonemes, <= 4 states, time_pruning <= 63, "
                                          "phonemes x states <= posterior outputs");
    c->dec_P = n_phonemes; c->dec_S = states_per_phn; c->dec_prune = time_pruning; c->dec_wpen = wpenalty;
    return LCRC_OK;
}

int lcrc_set_posterior_readback(lcrc_ctx *c, int enabled)
{
    if (!c) return LCRC_E_ARG;
    c->readback = enabled != 0;
    return LCRC_OK;
}

int lcrc_set_decoder_overlap(lcrc_ctx *c, int enabled)
{
    if (!c) return LCRC_E_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    if (c->dec_stream) HIP_TRY(c, hipStreamSynchronize(c->dec_stream));
    c->dec_pending = c->alt.dec_pending = false;
    if (enabled && !c->dec_stream) {
        HIP_TRY(c, hipStreamCreateWithFlags(&c->dec_stream, hipStreamNonBlocking));
        HIP_TRY(c, hipEventCreateWithFlags(&c->ev_post, hipEventDisableTiming));
        HIP_TRY(c, hipEventCreateWithFlags(&c->ev_dec_done, hipEventDisableTiming));
        HIP_TRY(c, hipEventCreateWithFlags(&c->alt.ev_dec_done, hipEventDisableTiming));
    }
    c->dec_overlap = enabled != 0;
    return LCRC_OK;
}

int lcrc_prev_labels(lcrc_ctx *c, const lcrc_label **labels, const int **first, const int **count, int *n_utts)
{
    if (!c || !labels || !first || !count || !n_utts) return LCRC_E_ARG;
    if (c->alt.dec_pending) {
        HIP_TRY(c, hipSetDevice(c->device));
        HIP_TRY(c, wait_event(c, c->alt.ev_dec_done));
        c->alt.dec_pending = false;
    }
    *labels = c->alt.h_labels;
    *first = c->alt.label_first.data();
    *count = c->alt.h_count;
    *n_utts = c->alt.label_utts;
    return LCRC_OK;
}

int lcrc_last_labels(lcrc_ctx *c, const lcrc_label **labels, const int **first, const int **count, int *n_utts)
{
    if (!c || !labels || !first || !count || !n_utts) return LCRC_E_ARG;
    if (c->dec_pending) {
        HIP_TRY(c, hipSetDevice(c->device));
        HIP_TRY(c, wait_event(c, c->ev_dec_done));
        c->dec_pending = false;
    }
    *labels = c->h_labels;
    *first = c->label_first.data();
    *count = c->h_count;
    *n_utts = c->label_utts;
    return LCRC_OK;
}

}  // extern "C"
