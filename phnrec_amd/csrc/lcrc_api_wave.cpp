// lcrc_api_wave.cpp -- the waveform entry points of the C ABI ("next" row f1; include/lcrc.h lcrc_frontend_configure ...):
// raw bytes in, the GPU mel-bank front-end (frontend_kernels.hip), sentence normalisation, then the posterior launch of
// lcrc_api.cpp; and lcrc_reserve, which allocates every buffer of these paths ahead of the first call.
#include "lcrc_ctx.h"

using namespace lcrc_impl;

extern "C" {

int lcrc_frontend_configure(lcrc_ctx *c, const lcrc_frontend *cfg)
{
    if (!c || !cfg) return LCRC_E_ARG;
    if ((cfg->wave_format != 1 && cfg->wave_format != 2) || cfg->vector_size < 2 || cfg->vector_size > 512 ||
        cfg->vector_step < 1 || cfg->sample_freq < 1)
        return fail(c, LCRC_E_ARG, "lcrc_frontend_configure: bad wave_format / vector_size (2..512) / vector_step / sample_freq");
    const int nbf = cfg->nbanks_full == -1 ? c->nbanks : cfg->nbanks_full;
    if (nbf < 3 || nbf < c->nbanks || nbf > 64)
        return fail(c, LCRC_E_ARG, "lcrc_frontend_configure: nbanks_full must be >= max(3, nbanks) and <= 64");
    HIP_TRY(c, hipSetDevice(c->device));
    const int fft = FftSizeFor(cfg->vector_size);
    if (fft != 256 && fft != 512) return fail(c, LCRC_E_UNSUPPORTED, "lcrc_frontend_configure: frames of 129..512 samples only (FFT 256 / 512)");
    std::vector<float> ham;
    BuildHamming(cfg->vector_size, ham);
    MelFilters mf;
    BuildMelFilters(nbf, fft, cfg->sample_freq, cfg->lower_freq, cfg->higher_freq, mf);
    std::vector<double> tw;
    BuildTwiddles(fft, tw);
    // contiguous runs of bins per bank: run 2b = bins with bank_of == b, run 2b+1 = bank_of == b+1
    std::vector<int> runs(4 * (size_t)nbf, 0);
    for (int b = 0; b < nbf; b++)
        for (int k = 0; k < 2; k++) {
            int lo = -1, hi = -1;
            for (int i = mf.fftlo; i <= mf.ffthi; i++)
                if (mf.bank_of[i] == b + k) { if (lo < 0) lo = i; hi = i + 1; }
            runs[2 * b + k] = lo < 0 ? 0 : lo;
            runs[2 * nbf + 2 * b + k] = lo < 0 ? 0 : hi;
        }
    for (void *p : {(void *)c->d_hamming, (void *)c->d_coeffs, (void *)c->d_twiddle, (void *)c->d_runs})
        if (p) (void)hipFree(p);
    c->d_hamming = c->d_coeffs = nullptr; c->d_twiddle = nullptr; c->d_runs = nullptr;
    HIP_TRY(c, hipMalloc((void **)&c->d_hamming, ham.size() * sizeof(float)));
    HIP_TRY(c, hipMalloc((void **)&c->d_coeffs, mf.coeffs.size() * sizeof(float)));
    HIP_TRY(c, hipMalloc((void **)&c->d_twiddle, tw.size() * sizeof(double)));
    HIP_TRY(c, hipMalloc((void **)&c->d_runs, runs.size() * sizeof(int)));
    // On the context's OWN stream, then one wait: a plain hipMemcpy runs on the device's null stream, whose queue the runtime
    // creates with its first use -- 8 ms and ~190 MB of resident memory (a queue's wave save area on this 256-CU device) that a
    // process which never touches the null stream never pays (PHNREC_TRACE_PIPELINE: "ctx: front-end set" 8.3 -> 0.3 ms).
    HIP_TRY(c, hipMemcpyAsync(c->d_hamming, ham.data(), ham.size() * sizeof(float), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(c->d_coeffs, mf.coeffs.data(), mf.coeffs.size() * sizeof(float), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(c->d_twiddle, tw.data(), tw.size() * sizeof(double), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(c->d_runs, runs.data(), runs.size() * sizeof(int), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));       // (the sources are this function's locals)
    c->fe = *cfg;
    c->fe.nbanks_full = nbf;
    c->fe_fft = fft;
    c->fe_ready = true;
    return LCRC_OK;
}

int lcrc_frontend_set_ln(lcrc_ctx *c, int form)
{
    if (!c) return LCRC_E_ARG;
    if (form < LCRC_LN_DOUBLE || form > LCRC_LN_GLIBC) return fail(c, LCRC_E_ARG, "lcrc_frontend_set_ln: form must be LCRC_LN_DOUBLE, LCRC_LN_GLIBC_FMA or LCRC_LN_GLIBC");
    c->fe_ln_form = form;
    return LCRC_OK;
}

int lcrc_device_ln(int device_id, int form, const float *x, float *y, long long n)
{
    if (n < 0 || (n > 0 && (!x || !y)) || form < LCRC_LN_DOUBLE || form > LCRC_LN_GLIBC) return fail(nullptr, LCRC_E_ARG, "lcrc_device_ln: bad argument");
    if (n == 0) return LCRC_OK;
    HIP_TRY(nullptr, hipSetDevice(device_id));
    float *d = nullptr;
    HIP_TRY(nullptr, hipMalloc((void **)&d, (size_t)n * sizeof(float)));
    hipError_t e = hipMemcpy(d, x, (size_t)n * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = frontend_ln_launch(d, (size_t)n, form, nullptr);
    if (e == hipSuccess) e = hipMemcpy(y, d, (size_t)n * sizeof(float), hipMemcpyDeviceToHost);
    (void)hipFree(d);
    HIP_TRY(nullptr, e);
    return LCRC_OK;
}

static long long fe_samples(const lcrc_ctx *c, long long n_bytes)
{
    return c->fe.wave_format == 1 ? n_bytes / 2 : n_bytes;
}

int lcrc_frontend_frames(const lcrc_ctx *c, long long n_bytes)
{
    if (!c || !c->fe_ready || n_bytes < 0) return LCRC_E_ARG;
    const long long len = fe_samples(c, n_bytes);
    return len > c->fe.vector_size ? (int)((len - c->fe.vector_size) / c->fe.vector_step + 1) : 1;
}

// LCRC_TRACE_SLOW_US=N (diagnostic): a waveform call that takes longer than N microseconds prints where it spent them
struct SlowTrace {
    long threshold_us;
    int n = 0;
    const char *what[12];
    std::chrono::steady_clock::time_point t[12];
    SlowTrace() { static const long th = getenv("LCRC_TRACE_SLOW_US") ? atol(getenv("LCRC_TRACE_SLOW_US")) : 0; threshold_us = th; mark("enter"); }
    void mark(const char *w) { if (threshold_us > 0 && n < 12) { what[n] = w; t[n++] = std::chrono::steady_clock::now(); } }
    ~SlowTrace()
    {
        if (threshold_us <= 0 || n < 2) return;
        const double total = std::chrono::duration<double, std::micro>(t[n - 1] - t[0]).count();
        if (total < (double)threshold_us) return;
        std::string line = "lcrc slow call at " + std::to_string((long long)std::chrono::duration<double, std::micro>(t[0].time_since_epoch()).count() % 100000000LL) +
                           " us, thread " + std::to_string((long)(size_t)pthread_self() % 1000) + " (" + std::to_string((long)total) + " us):";
        for (int i = 1; i < n; i++)
            line += std::string(" ") + what[i] + " +" + std::to_string((long)std::chrono::duration<double, std::micro>(t[i] - t[i - 1]).count());
        fprintf(stderr, "%s\n", line.c_str());
    }
};

// Shared by the two waveform entry points: stage the bytes (each utterance at an even offset),
// run the front-end into d_mel; on return *rows = total frames.
// Capacity of the pinned / device byte buffers of the waveform entry
static int ensure_wave_bytes(lcrc_ctx *c, long long total_bytes)
{
    if ((size_t)total_bytes + 16 > c->cap_bytes) {
        if (c->d_bytes) { (void)hipFree(c->d_bytes); (void)pinned_free(c->h_bytes); }
        c->d_bytes = c->h_bytes = nullptr; c->cap_bytes = 0;
        const size_t cap = (size_t)total_bytes + total_bytes / 4 + 4096;
        HIP_TRY(c, hipMalloc((void **)&c->d_bytes, cap));
        HIP_TRY(c, pinned_alloc((void **)&c->h_bytes, cap, true));      // (mapped: lcrc_wave_stage_energies pulls it by a kernel)
        c->cap_bytes = cap;
    }
    return LCRC_OK;
}

// per-utterance offsets and means of the waveform entry (2 * n_utts + 2 entries: sample starts and counts / frame and block offsets)
static int ensure_fe_utts(lcrc_ctx *c, size_t n_utts)
{
    if (2 * n_utts + 2 <= c->cap_fe_utts) return LCRC_OK;
    if (c->d_soff) { (void)hipFree(c->d_soff); (void)pinned_free(c->h_soff); (void)hipFree(c->d_foff); (void)pinned_free(c->h_foff); (void)hipFree(c->d_means); }
    c->d_soff = c->h_soff = nullptr; c->d_foff = c->h_foff = nullptr; c->d_means = nullptr; c->cap_fe_utts = 0;
    const size_t cap = 2 * n_utts + n_utts / 2 + 64;
    HIP_TRY(c, hipMalloc((void **)&c->d_soff, cap * sizeof(long long)));
    HIP_TRY(c, hipHostMalloc((void **)&c->h_soff, cap * sizeof(long long), kPinned));
    HIP_TRY(c, hipMalloc((void **)&c->d_foff, cap * sizeof(int)));
    HIP_TRY(c, hipHostMalloc((void **)&c->h_foff, cap * sizeof(int), kPinned));
    HIP_TRY(c, hipMalloc((void **)&c->d_means, cap * 64 * sizeof(float)));
    c->cap_fe_utts = cap;
    return LCRC_OK;
}

// partial sums of the tree mean (lcrc_set_mean_order(0)), one 64-float row per block of rows
static int ensure_mean_blocks(lcrc_ctx *c, size_t blocks)
{
    if (blocks <= c->cap_mean_blocks) return LCRC_OK;
    if (c->d_mean_part) (void)hipFree(c->d_mean_part);
    c->d_mean_part = nullptr; c->cap_mean_blocks = 0;
    const size_t cap = blocks + blocks / 4 + 64;
    HIP_TRY(c, hipMalloc((void **)&c->d_mean_part, cap * 64 * sizeof(float)));
    c->cap_mean_blocks = cap;
    return LCRC_OK;
}

// The front-end over utterances that already lie in the pinned byte buffer: utterance u = bytes
// [start[u], start[u] + len[u]) of c->h_bytes.  Leaves the features in c->d_mel.
static int run_frontend_staged(lcrc_ctx *c, const long long *start, const long long *len, int n_utts,
                               long long extent, int *frame_off, int *rows, bool raw_energies = false, SlowTrace *st = nullptr)
{
    long long total_frames = 0;
    for (int u = 0; u < n_utts; u++) total_frames += lcrc_frontend_frames(c, len[u]);
    if (total_frames > 0x7fffffffLL / 256) return fail(c, LCRC_E_ARG, "waveform entry: too many frames for one call");
    { const int rc = ensure_fe_utts(c, (size_t)n_utts); if (rc) return rc; }
    const int unit = c->fe.wave_format == 1 ? 2 : 1;
    c->h_foff[0] = 0;
    int *const h_boff = c->h_foff + n_utts + 1;      // block offsets of the tree mean, behind the frame offsets
    h_boff[0] = 0;
    for (int u = 0; u < n_utts; u++) {               // h_soff: [start of u ...][sample count of u ...]
        c->h_soff[u] = start[u] / unit;
        c->h_soff[n_utts + u] = fe_samples(c, len[u]);
        const int fr = lcrc_frontend_frames(c, len[u]);
        c->h_foff[u + 1] = c->h_foff[u] + fr;
        h_boff[u + 1] = h_boff[u] + meannorm_blocks(fr);
    }
    c->mean_blocks = h_boff[n_utts];
    { const int rc = ensure_mean_blocks(c, (size_t)c->mean_blocks); if (rc) return rc; }
    for (int u = 0; u < n_utts; u++) frame_off[u] = c->h_foff[u];
    if (n_utts >= 0) frame_off[n_utts] = c->h_foff[n_utts];
    *rows = (int)total_frames;
    if (total_frames == 0) return LCRC_OK;
    if (st) st->mark("offsets");
    int rc = ensure_staging(c, (size_t)total_frames, (size_t)n_utts);
    if (rc) return rc;
    if (st) st->mark("staging");
    // The energies entry keeps its two transfers out of the copy engine's queue, which every context of the device shares
    // in order: the samples are pulled by a kernel, the energies stored straight into the pinned feature buffer.  As copy
    // commands they stood behind other contexts' 24 MB of posteriors on their way back, which wait for those contexts'
    // kernels (-E: 24.7 M frames/s on one GPU with them).
    float *mel_out = c->d_mel;
    // (-E with the upload as a copy command and only the energies stored directly: 20.8-24.7 M; -F with its upload pulled
    //  by the kernel instead of copied: 27.4 against 28.7 M -- a caller that does not wait in between is better off with
    //  the copy engine; profiles/r04_ab_runs.txt 18)
    if (raw_energies) {
        void *src = nullptr;
        HIP_TRY(c, hipHostGetDevicePointer(&src, c->h_bytes, 0));
        HIP_TRY(c, pull_bytes_launch(src, c->d_bytes, (size_t)extent, c->stream));
        HIP_TRY(c, hipHostGetDevicePointer((void **)&mel_out, c->h_mel, 0));
    } else {
        HIP_TRY(c, hipMemcpyAsync(c->d_bytes, c->h_bytes, (size_t)extent, hipMemcpyHostToDevice, c->stream));
    }
    if (st) st->mark("bytes copy queued");
    HIP_TRY(c, hipMemcpyAsync(c->d_soff, c->h_soff, (size_t)(2 * n_utts) * sizeof(long long), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(c->d_foff, c->h_foff, (size_t)(2 * n_utts + 2) * sizeof(int), hipMemcpyHostToDevice, c->stream));
    if (st) st->mark("offset copies queued");
    FrontendParams p;
    memset(&p, 0, sizeof p);
    p.bytes = c->d_bytes; p.sample_start = c->d_soff; p.frame_off = c->d_foff; p.mel = mel_out;
    p.hamming = c->d_hamming; p.twiddle = c->d_twiddle; p.coeffs = c->d_coeffs;
    p.run_begin = c->d_runs; p.run_end = c->d_runs + 2 * c->fe.nbanks_full;
    p.n_utts = n_utts; p.n_frames = (int)total_frames; p.nbanks = c->nbanks; p.fft = c->fe_fft;
    p.wave_format = c->fe.wave_format; p.vector_size = c->fe.vector_size; p.vector_step = c->fe.vector_step;
    p.dc_shift = c->fe.dc_shift; p.scale = c->fe.scale; p.preem_coef = c->fe.preem_coef;
    p.z_mean_source = c->fe.z_mean_source;
    p.raw_energies = raw_energies ? 1 : 0;
    p.ln_form = c->fe_ln_form;
    HIP_TRY(c, frontend_launch(p, c->stream));
    return LCRC_OK;
}

static int run_frontend(lcrc_ctx *c, const unsigned char *bytes, const long long *byte_off, int n_utts,
                        int *frame_off, int *rows)
{
    if (!c->fe_ready) return fail(c, LCRC_E_ARG, "waveform entry used before lcrc_frontend_configure");
    if (n_utts < 0 || (n_utts > 0 && (!bytes || !byte_off || !frame_off)) || (n_utts > 0 && byte_off[0] != 0))
        return fail(c, LCRC_E_ARG, "waveform entry: bad argument");
    HIP_TRY(c, hipSetDevice(c->device));
    long long total_bytes = 0;
    for (int u = 0; u < n_utts; u++) {
        const long long nb = byte_off[u + 1] - byte_off[u];
        if (nb < 0) return fail(c, LCRC_E_ARG, "waveform entry: offsets must be non-decreasing");
        total_bytes += nb + (nb & 1);
    }
    int rc = ensure_wave_bytes(c, total_bytes);
    if (rc) return rc;
    std::vector<long long> start((size_t)std::max(n_utts, 0)), len((size_t)std::max(n_utts, 0));
    long long pos = 0;
    for (int u = 0; u < n_utts; u++) {
        const long long nb = byte_off[u + 1] - byte_off[u];
        memcpy(c->h_bytes + pos, bytes + byte_off[u], (size_t)nb);
        start[u] = pos; len[u] = nb;
        pos += nb + (nb & 1);                        // keep lin16 utterances 2-byte aligned
    }
    return run_frontend_staged(c, start.data(), len.data(), n_utts, pos, frame_off, rows);
}

int lcrc_wave_to_mel(lcrc_ctx *c, const unsigned char *bytes, const long long *byte_off, int n_utts,
                     float *mel, int *frame_off)
{
    if (!c) return LCRC_E_ARG;
    int rows = 0;
    int rc = run_frontend(c, bytes, byte_off, n_utts, frame_off, &rows);
    if (rc || rows == 0) return rc;
    if (!mel) return fail(c, LCRC_E_ARG, "lcrc_wave_to_mel: NULL output");
    const size_t nbytes = (size_t)rows * c->nbanks * sizeof(float);
    HIP_TRY(c, hipMemcpyAsync(c->h_mel, c->d_mel, nbytes, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, wait_stream(c));
    memcpy(mel, c->h_mel, nbytes);
    return LCRC_OK;
}

// the part of the waveform -> posteriors entries behind the front-end
static int wave_finish(lcrc_ctx *c, int n_utts, int rows, float *post, SlowTrace *st = nullptr, bool staged = false)
{
    const bool copy_post = c->readback || c->dec_P <= 0;
    if (c->fe.sent_mean_norm) {
        int longest = 0;                         // (h_foff: the host copy of this call's frame offsets)
        for (int u = 0; u < n_utts; u++) longest = std::max(longest, c->h_foff[u + 1] - c->h_foff[u]);
        HIP_TRY(c, meannorm_launch(c->d_mel, c->d_foff, c->mean_sequential ? nullptr : c->d_foff + n_utts + 1, c->mean_blocks,
                                   c->d_mean_part, n_utts, rows, c->nbanks, c->d_means, longest, c->stream));
    }
    if (st) st->mark("mean queued");
    if (copy_post && c->dec_P <= 0) {
        bool done = false;
        const int rc2 = two_part_output(c, c->d_mel, c->d_foff, n_utts, rows, post, &done);
        if (rc2 || done) { if (!rc2) c->label_utts = 0; return rc2; }
    }
    float *out_dev = c->d_post;
    bool direct = false;
    int rc = output_target(c, copy_post, post == nullptr, &out_dev, &direct);
    if (rc) return rc;
    rc = launch(c, c->d_mel, c->d_foff, n_utts, rows, out_dev, c->stream, nullptr);
    if (rc) return rc;
    if (st) st->mark("kernels queued");
    rc = decode_after(c, c->d_foff, c->h_foff, n_utts, rows, c->d_post, c->stream, staged);
    if (rc) return rc;
    const size_t nbytes = (size_t)rows * c->nets[2].n_out * sizeof(float);
    if (copy_post && direct) {
        HIP_TRY(c, wait_stream(c));
        if (post) memcpy(post, c->h_post, nbytes);
    } else if (copy_post) {
        rc = ensure_host_post(c);
        if (rc) return rc;
        if (st) st->mark("host buffer");
        HIP_TRY(c, copy_back(c, post, c->h_post, c->d_post, nbytes));   // post == NULL: read them in place (lcrc_staged_posteriors)
    } else {
        HIP_TRY(c, wait_stream(c));
    }
    if (st) st->mark("done");
    return LCRC_OK;
}

int lcrc_staged_posteriors(lcrc_ctx *c, const float **post)
{
    if (!c || !post) return LCRC_E_ARG;
    *post = c->h_post;
    return LCRC_OK;
}

int lcrc_wave_to_posteriors(lcrc_ctx *c, const unsigned char *bytes, const long long *byte_off, int n_utts,
                            float *post, int *frame_off)
{
    if (!c) return LCRC_E_ARG;
    int rows = 0;
    int rc = run_frontend(c, bytes, byte_off, n_utts, frame_off, &rows);
    if (rc || rows == 0) { c->label_utts = 0; return rc; }
    if (!post && (c->readback || c->dec_P <= 0)) return fail(c, LCRC_E_ARG, "lcrc_wave_to_posteriors: NULL output");
    return wave_finish(c, n_utts, rows, post);
}

int lcrc_wave_stage_buffer(lcrc_ctx *c, long long capacity, unsigned char **bytes)
{
    if (!c || capacity < 0 || !bytes) return fail(c, LCRC_E_ARG, "lcrc_wave_stage_buffer: bad argument");
    HIP_TRY(c, hipSetDevice(c->device));
    int rc = ensure_wave_bytes(c, capacity);
    if (rc) return rc;
    *bytes = c->h_bytes;
    return LCRC_OK;
}

// Every buffer a later call of up to max_rows frames in max_utts utterances (and max_wave_bytes of waveform, 0: the
// frame entries only) would allocate on demand, allocated now: device staging, pinned features and posteriors, byte
// buffers, per-utterance offsets.  What on-demand growth costs is page pinning -- 744 B per HU frame, ~8 ms per
// 32 768 frames -- inside the first call; a caller with a warm-up phase (the CLI, while its list is still being opened)
// pays it there, all contexts at once.
int lcrc_reserve(lcrc_ctx *c, int max_rows, int max_utts, long long max_wave_bytes)
{
    if (!c) return LCRC_E_ARG;
    if (max_rows < 0 || max_utts < 0 || max_wave_bytes < 0) return fail(c, LCRC_E_ARG, "lcrc_reserve: negative size");
    if (max_wave_bytes > 0 && !c->fe_ready) return fail(c, LCRC_E_ARG, "lcrc_reserve: waveform bytes asked for before lcrc_frontend_configure");
    HIP_TRY(c, hipSetDevice(c->device));
    int rc = ensure_staging(c, (size_t)std::max(max_rows, 1), (size_t)max_utts);
    if (rc) return rc;
    if (c->readback || c->dec_P <= 0) {
        rc = ensure_host_post(c);
        if (rc) return rc;
    }
    if (c->system == SYS_LCRC && !c->d_part && c->split_hint != 1 && c->arith == 0) ensure_split_scratch(c);
    if (c->dec_P > 0) {
        rc = ensure_labels(c, (size_t)std::max(max_rows, 1), (size_t)max_utts);
        if (rc) return rc;
        if (overlap_on(c)) {                     // the second set: posterior buffer and labels
            swap_decoder_sets(c);
            rc = ensure_post_rows(c);
            if (!rc) rc = ensure_labels(c, (size_t)std::max(max_rows, 1), (size_t)max_utts);
            swap_decoder_sets(c);
            if (rc) return rc;
        }
    }
    if (max_wave_bytes > 0) {
        rc = ensure_wave_bytes(c, max_wave_bytes);
        if (rc) return rc;
        rc = ensure_fe_utts(c, (size_t)max_utts);
        if (rc) return rc;
        rc = ensure_mean_blocks(c, (size_t)meannorm_blocks(max_rows) + (size_t)max_utts);
        if (rc) return rc;
    }
    // The copy engines.  The runtime creates an SDMA queue the first time it uses an engine (~13 ms each, startup_probe:
    // "first copy"), and picks a further engine whenever the ones it has are busy -- which, with several contexts copying
    // bytes in and posteriors out at once, happened twice in the first 40 ms of every list, EVERY copy of the process waiting
    // meanwhile (tools/pipeline_trace.py: hipMemcpyAsync blocking 13-15 ms).  So the buffers make a few round trips now, in
    // both directions at once; contexts reserved from parallel threads overlap the way a list's launches will.
    if (max_rows >= 4096) {
        const size_t post_bytes = (size_t)max_rows * c->nets[2].n_out * sizeof(float);
        const size_t mel_bytes = (size_t)max_rows * c->nbanks * sizeof(float);
        for (int round = 0; round < 3; round++) {
            if (max_wave_bytes > 0) HIP_TRY(c, hipMemcpyAsync(c->d_bytes, c->h_bytes, (size_t)max_wave_bytes, hipMemcpyHostToDevice, c->stream));
            else HIP_TRY(c, hipMemcpyAsync(c->d_mel, c->h_mel, mel_bytes, hipMemcpyHostToDevice, c->stream));
            if (c->h_post) {                                   // in pieces, the way copy_back() queues them
                const size_t step = ((post_bytes / kCopyPieces) + 255) & ~(size_t)255;
                for (int k = 0; k < kCopyPieces; k++) {
                    const size_t lo = (size_t)k * step, n = k + 1 == kCopyPieces ? post_bytes - lo : step;
                    HIP_TRY(c, hipMemcpyAsync((char *)c->h_post + lo, (const char *)c->d_post + lo, n, hipMemcpyDeviceToHost, c->stream));
                }
            }
        }
        HIP_TRY(c, hipStreamSynchronize(c->stream));
    }
    return LCRC_OK;
}

int lcrc_wave_stage_energies(lcrc_ctx *c, const long long *start, const long long *n_bytes, int n_utts, float **energies,
                             int *frame_off)
{
    if (!c) return LCRC_E_ARG;
    if (!c->fe_ready) return fail(c, LCRC_E_ARG, "waveform entry used before lcrc_frontend_configure");
    if (n_utts < 0 || !energies || (n_utts > 0 && (!start || !n_bytes || !frame_off)))
        return fail(c, LCRC_E_ARG, "lcrc_wave_stage_energies: bad argument");
    *energies = nullptr;
    HIP_TRY(c, hipSetDevice(c->device));
    long long extent = 0;
    for (int u = 0; u < n_utts; u++) {
        if (start[u] < extent || n_bytes[u] < 0 || (c->fe.wave_format == 1 && (start[u] & 1)))
            return fail(c, LCRC_E_ARG, "lcrc_wave_stage_energies: utterances must be in order, not overlap, and start on even bytes (lin16)");
        extent = start[u] + n_bytes[u];
    }
    if ((size_t)extent + 16 > c->cap_bytes) return fail(c, LCRC_E_ARG, "lcrc_wave_stage_energies: beyond the capacity lcrc_wave_stage_buffer reserved");
    int rows = 0;
    int rc = run_frontend_staged(c, start, n_bytes, n_utts, extent, frame_off, &rows, true);
    if (rc || rows == 0) return rc;
    rc = ensure_host_post(c);                    // lcrc_stage_run follows: everything it needs exists now and will not move
    if (rc) return rc;
    c->kdone_armed = false;
    HIP_TRY(c, wait_stream(c));                  // (the kernel has stored the energies in the pinned buffer itself)
    *energies = c->h_mel;
    return LCRC_OK;
}

int lcrc_wave_stage_run(lcrc_ctx *c, const long long *start, const long long *n_bytes, int n_utts, float *post,
                        int *frame_off)
{
    if (!c) return LCRC_E_ARG;
    if (!c->fe_ready) return fail(c, LCRC_E_ARG, "waveform entry used before lcrc_frontend_configure");
    if (n_utts < 0 || (n_utts > 0 && (!start || !n_bytes || !frame_off))) return fail(c, LCRC_E_ARG, "lcrc_wave_stage_run: bad argument");
    HIP_TRY(c, hipSetDevice(c->device));
    long long extent = 0;
    for (int u = 0; u < n_utts; u++) {
        if (start[u] < extent || n_bytes[u] < 0 || (c->fe.wave_format == 1 && (start[u] & 1)))
            return fail(c, LCRC_E_ARG, "lcrc_wave_stage_run: utterances must be in order, not overlap, and start on even bytes (lin16)");
        extent = start[u] + n_bytes[u];
    }
    if ((size_t)extent + 16 > c->cap_bytes) return fail(c, LCRC_E_ARG, "lcrc_wave_stage_run: beyond the capacity lcrc_wave_stage_buffer reserved");
    int rows = 0;
    SlowTrace st;
    OverlapScope scope(c);
    int rc = post ? settle_pending_decoders(c) : begin_overlapped_call(c);
    if (rc) return rc;
    rc = run_frontend_staged(c, start, n_bytes, n_utts, extent, frame_off, &rows, false, &st);
    if (rc || rows == 0) { c->label_utts = 0; return rc; }
    if (!post && overlap_on(c)) { rc = ensure_post_rows(c); if (rc) return rc; }      // (the staging may just have grown)
    st.mark("front-end queued");
    return wave_finish(c, n_utts, rows, post, &st, post == nullptr);
}

}  // extern "C"
