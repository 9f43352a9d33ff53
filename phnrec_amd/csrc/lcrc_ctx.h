// lcrc_ctx.h -- what the translation units behind include/lcrc.h share: the context (struct lcrc_ctx), the model its clones
// share, and the helpers that cross files.  lcrc_api.cpp: model loading / packing, contexts, staging, the posterior launches
// and the frame entry points; lcrc_api_wave.cpp: the waveform entry points (GPU front-end) and lcrc_reserve;
// lcrc_api_decoder.cpp: the decoder on the device (label buffers, launch behind the posterior kernel, overlap mode).
#ifndef PHNREC_LCRC_CTX_H
#define PHNREC_LCRC_CTX_H
#include "../../include/lcrc_experimental.h"      // (includes lcrc_pipeline.h and lcrc.h: the library defines them all)

#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <pthread.h>
#include <string>
#include <thread>
#include <time.h>
#include <vector>

#include "frontend_dev.h"
#include "lcrc_dev.h"
#include "meltables.h"
#include "nnet_io.h"

using namespace phnrec;

// SYS_LCRC: the fused kernels (length 31, add_c0, 11 coefficients per band: every shipped model); SYS_LCRC_GEN: LCRC at any
// other geometry the reference accepts, composed from the general features / MLP kernels like the unfused other systems
enum { SYS_LCRC = 0, SYS_1BT_DCT = 1, SYS_1BT = 2, SYS_3BT = 3, SYS_LCRC_GEN = 4 };

// split-f16 operand images of one net (pack_net_h2)
struct H2Images {
    const float4 *w1h = nullptr, *w2h = nullptr;
    const float *b1h = nullptr, *b2h = nullptr;
    float sig_descale = 1.f, out_descale = 1.f;
};

// What the contexts of one model on one GPU share (lcrc_clone): the read-only device buffers -- packed weights,
// biases, normalisation vectors, tables -- and the host copy of the nets the split-f16 operand images are packed
// from on first request.  Freed with the last context.
struct SharedModel {
    int device = 0;
    std::vector<void *> allocs;
    HostNet host[3];
    std::mutex mu;                       // guards the lazily built split-f16 images
    int h2_state = 0;                    // 0: not built, 1: built, -1: the model has no such form
    H2Images h2[3];
    ~SharedModel()
    {
        if (allocs.empty()) return;
        (void)hipSetDevice(device);
        for (void *p : allocs) (void)hipFree(p);
    }
};

struct lcrc_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    int nbanks = 0;
    std::shared_ptr<SharedModel> model;
    NetDev nets[3];
    std::vector<void *> allocs;          // this context's own device buffers (split scratch)
    float *d_win = nullptr, *d_costab = nullptr;
    float normc = 0.f;
    // the other posteriors/system variants ("next" row f4): nets[2] is the merger in every system
    int system = SYS_LCRC, trap_bands = 0, shift = 0;
    bool use_hamming = false, add_c0 = true;
    std::vector<NetDev> band_nets;       // 1BT / 3BT: trap_bands nets of 31 inputs
    const NetDev *d_band_nets = nullptr; // the same on the device (one launch runs them all)
    const int *d_band_col = nullptr;     // first merger-input column of each band net
    NetDev band_max = {};                // maxima of ksteps / nkq / n_ot over the band nets
    float *d_hamm31 = nullptr, *d_costab31 = nullptr;    // (named for the usual length; sized by trap_len)
    float normc31 = 0.f;
    int trap_len = kTrapLen;             // posteriors/length; anything but 31 runs the general (unfused) kernels
    float *d_win_gen = nullptr;          // SYS_LCRC_GEN: [2][half] windows
    float *d_feat = nullptr, *d_minp = nullptr;   // trajectories or C0/DCT rows; merger input of 1BT / 3BT
    size_t cap_feat_rows = 0;
    bool traps_unfused = false;          // PHNREC_TRAPS_UNFUSED=1: every system as separate features / MLP launches (A/B, tests)
    bool bt_unfused = false;             // 1BT / 3BT model that no fused size class holds
    const char *mlp_variant = "none";    // kernel of the last merger launch (1BT_DCT / 1BT / 3BT)
    // staging for the host-pointer entry points (grown on demand)
    float *d_mel = nullptr, *d_post = nullptr;
    int *d_off = nullptr;
    float *h_mel = nullptr, *h_post = nullptr;
    int *h_off = nullptr;
    size_t cap_rows = 0, cap_utts = 0, cap_host_post = 0;
    float *d_dbg[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    size_t cap_dbg = 0;
    // GPU front-end ("next" row f1): configuration, device tables, staging for raw bytes
    bool fe_ready = false;
    lcrc_frontend fe = {};
    int fe_fft = 0;
    int fe_ln_form = 0;                  // LCRC_LN_* (lcrc_frontend_set_ln)
    float *d_hamming = nullptr, *d_coeffs = nullptr;
    double *d_twiddle = nullptr;
    int *d_runs = nullptr;               // [4*nbanks_full]: run_begin[2*nbf], run_end[2*nbf]
    unsigned char *d_bytes = nullptr, *h_bytes = nullptr;
    long long *d_soff = nullptr, *h_soff = nullptr;
    int *d_foff = nullptr, *h_foff = nullptr;
    float *d_means = nullptr, *d_mean_part = nullptr;
    size_t cap_bytes = 0, cap_fe_utts = 0, cap_mean_blocks = 0;
    int mean_blocks = 0;                 // blocks of the last staged batch (tree mean)
    bool mean_sequential = true;         // lcrc_set_mean_order: the reference's order unless the caller opts out
    // streaming state (lcrc_push): the pushed frames live in a pinned, device-mapped strip whose last 30 rows
    // are the history (Traps::be_mat minus its newest slot); the kernel reads the strip and writes the
    // posteriors of a push in place (zero-copy), so a push costs no allocation and no copy command
    float *h_ring = nullptr, *d_ring = nullptr;       // [ring_cap][nbanks], host and device view
    float *h_pushout = nullptr, *d_pushout = nullptr; // [pushout_cap][n_out]
    size_t ring_cap = 0, ring_rows = 0, pushout_cap = 0;
    bool hist_init = false;
    int delay = 0;
    // split-hidden path (small launches): scratch for partial output tiles, operand images, tickets
    float4 *d_part = nullptr, *d_gimg = nullptr;
    unsigned *d_cnt = nullptr;
    int split_hint = 0;
    bool split_scratch_failed = false;
    // timing
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool timing = true, timed = false;
    int poll_wait_us = 0;                // lcrc_set_wait_mode: 0 = spin in hipStreamSynchronize, > 0 = sleep between completion queries
    hipEvent_t ev_wait = nullptr;
    hipEvent_t ev_piece[8] = {};         // copy_back's pieces
    lcrc_kernel_done_fn kdone_fn = nullptr;      // lcrc_set_kernel_done_callback
    void *kdone_arg = nullptr;
    hipEvent_t ev_kdone = nullptr;
    bool kdone_armed = false;            // an event behind this call's posterior kernels is recorded and not yet reported
    // posterior writer path
    lcrc_softening soft[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
    int out_be = 0;
    int tile_frames = 0;
    int arith = 0;              // LCRC_ARITH_*
    // decoder on the device ("next" row f3): configuration, label buffers (device + pinned host)
    int dec_P = 0, dec_S = 0, dec_prune = 0;
    float dec_wpen = 0.f;
    bool readback = true;
    lcrc_label *d_labels = nullptr, *h_labels = nullptr;
    int *d_count = nullptr, *h_count = nullptr;
    size_t cap_label_rows = 0, cap_label_utts = 0;
    std::vector<int> label_first;
    int label_utts = 0;
    // lcrc_set_decoder_overlap: the decoder kernel of a staged call runs on dec_stream, behind an event,
    // BESIDE the next call's front-end and posterior kernels.  Everything a launch's decoder reads or writes exists twice and
    // alternates: the context's own fields above (d_post, d_labels ... label_utts, with d_dec_off / ev_dec_done / dec_pending)
    // are the set of the CURRENT call, `alt` is the set of the call before it.
    bool launch_ordered = false;         // lcrc_set_launch_order: posterior kernels of this device's ordered contexts run one after the other
    bool dec_overlap = false;
    hipStream_t dec_stream = nullptr;
    hipEvent_t ev_post = nullptr;        // this call's posterior kernels (and the decoder's copy of the offsets) are done
    size_t d_post_cap = 0;               // rows d_post holds (cap_rows unless the sets have just been swapped)
    int *d_dec_off = nullptr;            // the decoder's own copy of the utterance offsets (the next call overwrites d_off / d_foff)
    size_t cap_dec_off = 0;
    hipEvent_t ev_dec_done = nullptr;    // behind the decoder kernel of this set's last launch
    bool dec_pending = false;            // ... recorded and not yet waited for
    bool overlapped_call = false;        // inside a staged call that takes the overlapped path (begin_overlapped_call)
    struct DecSet {
        float *d_post = nullptr;
        size_t d_post_cap = 0;
        lcrc_label *d_labels = nullptr, *h_labels = nullptr;
        int *d_count = nullptr, *h_count = nullptr;
        size_t cap_label_rows = 0, cap_label_utts = 0;
        std::vector<int> label_first;
        int label_utts = 0;
        int *d_dec_off = nullptr;
        size_t cap_dec_off = 0;
        hipEvent_t ev_dec_done = nullptr;
        bool dec_pending = false;
    } alt;
    unsigned long long *d_stamps = nullptr;   // diagnostic build only
    std::string err;
    const char *variant = "none";
    unsigned lds_bytes = 0;
};

namespace lcrc_impl {

// error text into the context (or, without one, into the thread's creation error) and `code` back
int fail(lcrc_ctx *c, int code, const std::string &msg);

#define HIP_TRY(ctx, expr)                                                                  \
    do {                                                                                    \
        hipError_t e_ = (expr);                                                             \
        if (e_ != hipSuccess)                                                               \
            return fail(ctx, LCRC_E_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

constexpr unsigned kPinned = hipHostMallocPortable;
constexpr unsigned kPinnedMapped = hipHostMallocPortable | hipHostMallocMapped | hipHostMallocCoherent;
constexpr int kCopyPieces = 4;           // <= lcrc_ctx::ev_piece

// ---- lcrc_api.cpp ----
hipError_t dev_alloc(void **p, size_t bytes);
hipError_t pinned_alloc(void **p, size_t bytes, bool device_reads = false);
hipError_t pinned_free(void *p);         // for anything pinned_alloc or hipHostMalloc handed out
hipError_t wait_stream(lcrc_ctx *c);
hipError_t wait_event(lcrc_ctx *c, hipEvent_t ev);
hipError_t copy_back(lcrc_ctx *c, float *dst, float *pinned, const float *dev, size_t nbytes);
int ensure_staging(lcrc_ctx *c, size_t rows, size_t utts);
int ensure_host_post(lcrc_ctx *c);
int output_target(lcrc_ctx *c, bool copy_post, bool in_place, float **out, bool *direct);
void ensure_split_scratch(lcrc_ctx *c);
int launch(lcrc_ctx *c, const float *d_mel, const int *d_off, int n_utts, int n_rows, float *d_post,
           hipStream_t s, float *const *dbg, int row_first = 0, int row_count = -1, bool timed = true);
int two_part_output(lcrc_ctx *c, const float *d_mel, const int *d_off, int n_utts, int n, float *post, bool *done);

// ---- lcrc_api_decoder.cpp ----
int ensure_labels(lcrc_ctx *c, size_t n_rows, size_t n_utts);
bool overlap_on(const lcrc_ctx *c);
void swap_decoder_sets(lcrc_ctx *c);
int ensure_post_rows(lcrc_ctx *c);
int begin_overlapped_call(lcrc_ctx *c);
int settle_pending_decoders(lcrc_ctx *c);
// a staged entry point's stay: whatever begin_overlapped_call decided ends with the call
struct OverlapScope {
    lcrc_ctx *c;
    explicit OverlapScope(lcrc_ctx *x) : c(x) {}
    ~OverlapScope() { if (c) c->overlapped_call = false; }
};
int decode_after(lcrc_ctx *c, const int *d_off, const int *h_first, int n_utts, int n_rows, const float *d_post,
                 hipStream_t s, bool staged = false);

}  // namespace lcrc_impl

#endif
