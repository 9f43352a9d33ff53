// lcrc_api.cpp -- the C ABI of include/lcrc.h over the fused HIP kernel.
//
// Host side of the drop-in boundary: loads a PhnRec model directory with the
// reference's file formats (nnet_io.cpp), re-packs the three MLPs into MFMA
// fragment order, owns the device buffers / stream / events, and mirrors the
// two ways SpeechRec drives Traps: whole utterances (srec.cpp:1035-1059) and the
// streaming CalcFeaturesBunched form (traps.cpp:518-535).
// There is deliberately no CPU path in this file.  (The waveform entry points: lcrc_api_wave.cpp; the decoder on the
// device: lcrc_api_decoder.cpp; what the three share: lcrc_ctx.h.)
#include "lcrc_ctx.h"

#include <sys/mman.h>

#include <cstdint>
#include <unordered_map>

namespace lcrc_impl {

thread_local std::string g_create_err = "";

// LCRC_TRACE_STARTUP=1: where a context's creation spends its time, one line per phase on stderr (diagnostic)
struct StartupTrace {
    bool on;
    std::chrono::steady_clock::time_point t;
    StartupTrace() : on(getenv("LCRC_TRACE_STARTUP") != nullptr), t(std::chrono::steady_clock::now()) {}
    void mark(const char *what)
    {
        if (!on) return;
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "lcrc startup: %-28s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(now - t).count());
        t = now;
    }
};

int fail(lcrc_ctx *c, int code, const std::string &msg)
{
    if (c) c->err = msg; else g_create_err = msg;
    return code;
}


// (the copy is queued on the context's stream: a synchronous hipMemcpy from pageable memory costs a fresh process twice as
//  much -- 17 ms against 8 ms for the first 6.6 MB, tools/ubench/hip_upload -- and every later use of the buffer is
//  ordered behind it on that stream anyway; the creating call synchronises once before it hands the context out)
template <typename T>
hipError_t dev_upload(lcrc_ctx *c, const std::vector<T> &h, const T **out)
{
    void *d = nullptr;
    hipError_t e = hipMalloc(&d, h.size() * sizeof(T));
    if (e != hipSuccess) return e;
    c->model->allocs.push_back(d);
    e = hipMemcpyAsync(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);      // pageable source: staged by the call; the wait is short
    *out = static_cast<const T *>(d);
    return e;
}

// Split-f16 form of a net (mlp_dev.h HalfLoop): every weight as a (high, low) f16 pair, high = f16(w'), low =
// f16(w' - high), w' = w * 2^e, in the A-fragment order of v_mfma_f32_16x16x32_f16 (lane l: row l&15, k-slots 8(l>>4) .. +7):
//   w1h[(((P*ns + s)*2 + T)*2 + piece)*64 + l][j] = W1[32P + 16T + (l&15)][32s + 8(l>>4) + j]
//   w2h[((P*n_ot + ot)*2 + piece)*64 + l][j]      = W2[16ot + (l&15)][32P + (j < 4 ? 4(l>>4) + j : 16 + 4(l>>4) + j - 4)]
// (layer 2's k-slots follow the accumulator layout of the pair's two layer-1 tiles).
// The exponent e of a matrix puts its largest weight in (2^13, 2^14]: the low half of a value below 2^-3 would be an
// f16 subnormal (2^-25 absolute instead of 2^-22 relative -- a model with weights of 1e-3 would lose ten bits), and
// a weight beyond f16's range comes back into it.  The biases are stored in the accumulators' scale; the kernels undo
// the scaling exactly (powers of two).  A non-finite weight has no such form.
int h2_exponent(const std::vector<float> &w, bool *finite)
{
    float m = 0.f;
    for (float v : w) {
        if (!std::isfinite(v)) { *finite = false; return 0; }
        m = std::max(m, fabsf(v));
    }
    if (m == 0.f) return 0;
    int ex = 0;
    (void)frexpf(m, &ex);                        // m = f * 2^ex, f in [0.5, 1)
    return std::max(-60, std::min(60, 14 - ex));  // m * 2^e in [2^13, 2^14)
}

int pack_net_h2(lcrc_ctx *c, const HostNet &h, const NetDev &d, H2Images &out, bool *representable)
{
    *representable = true;
    const int e1 = h2_exponent(h.w1, representable), e2 = h2_exponent(h.w2, representable);
    if (!*representable) return LCRC_OK;
    const int ns = (h.n_inp + 31) / 32;
    const float s1 = ldexpf(1.0f, e1), s2 = ldexpf(1.0f, e2);
    std::vector<_Float16> w1h((size_t)d.npairs * ns * 2 * 2 * 64 * 8, (_Float16)0.0f);
    std::vector<_Float16> w2h((size_t)d.npairs * d.n_ot * 2 * 64 * 8, (_Float16)0.0f);
    auto put = [](std::vector<_Float16> &a, size_t frag, int l, int j, float w) {
        const _Float16 hi = (_Float16)w;
        a[(frag * 64 + l) * 8 + j] = hi;
        a[((frag + 1) * 64 + l) * 8 + j] = (_Float16)(w - (float)hi);
    };
    for (int P = 0; P < d.npairs; P++)
        for (int s = 0; s < ns; s++)
            for (int T = 0; T < 2; T++)
                for (int l = 0; l < 64; l++)
                    for (int j = 0; j < 8; j++) {
                        const int hh = 32 * P + 16 * T + (l & 15), k = 32 * s + 8 * (l >> 4) + j;
                        if (hh < h.n_hid && k < h.n_inp)
                            put(w1h, (((size_t)P * ns + s) * 2 + T) * 2, l, j, h.w1[(size_t)hh * h.n_inp + k] * s1);
                    }
    for (int P = 0; P < d.npairs; P++)
        for (int ot = 0; ot < d.n_ot; ot++)
            for (int l = 0; l < 64; l++)
                for (int j = 0; j < 8; j++) {
                    const int g = l >> 4, o = 16 * ot + (l & 15);
                    const int hh = 32 * P + (j < 4 ? 4 * g + j : 16 + 4 * g + j - 4);
                    if (o < h.n_out && hh < h.n_hid)
                        put(w2h, ((size_t)P * d.n_ot + ot) * 2, l, j, h.w2[(size_t)o * h.n_hid + hh] * s2);
                }
    // biases in the accumulators' scale: inputs carry 2^6 (kH2InScale), activations 2^14 (kH2ActScale)
    const float a1 = ldexpf(1.0f, e1 + 6), a2 = ldexpf(1.0f, e2 + 14);
    std::vector<float> b1h((size_t)d.npairs * 32, 0.f), b2h((size_t)d.n_ot * 16, 0.f);
    for (int i = 0; i < h.n_hid; i++) b1h[i] = h.b1[i] * a1;
    for (int i = 0; i < h.n_out; i++) b2h[i] = h.b2[i] * a2;
    for (float v : b1h) if (!std::isfinite(v)) *representable = false;
    for (float v : b2h) if (!std::isfinite(v)) *representable = false;
    if (!*representable) return LCRC_OK;
    const _Float16 *p = nullptr;
    HIP_TRY(c, dev_upload(c, w1h, &p)); out.w1h = reinterpret_cast<const float4 *>(p);
    HIP_TRY(c, dev_upload(c, w2h, &p)); out.w2h = reinterpret_cast<const float4 *>(p);
    HIP_TRY(c, dev_upload(c, b1h, &out.b1h));
    HIP_TRY(c, dev_upload(c, b2h, &out.b2h));
    out.sig_descale = ldexpf(1.0f, -(e1 + 6));
    out.out_descale = ldexpf(1.0f, -(e2 + 14));
    return LCRC_OK;
}

// The split-f16 operand images of a model: built and uploaded on the FIRST lcrc_set_arithmetic(.., LCRC_ARITH_SPLIT_F16)
// of any of the contexts that share the model (a context that never asks pays nothing), then handed to this context.
int ensure_split_f16(lcrc_ctx *c)
{
    SharedModel &m = *c->model;
    std::lock_guard<std::mutex> l(m.mu);
    if (m.h2_state == 0) {
        // -1 (no such form, for good) only when a net is not representable; a device error (allocation, upload) leaves
        // the state at 0 so that a later call retries -- the images uploaded so far stay in the model's allocation list
        // (freed with the model) and are simply re-made
        bool ok = true;
        for (int i = 0; i < 3 && ok; i++) {
            int rc = pack_net_h2(c, m.host[i], c->nets[i], m.h2[i], &ok);
            if (rc) return rc;
        }
        m.h2_state = ok ? 1 : -1;
    }
    if (m.h2_state != 1) return LCRC_E_UNSUPPORTED;
    for (int i = 0; i < 3; i++) {
        c->nets[i].w1h = m.h2[i].w1h; c->nets[i].w2h = m.h2[i].w2h;
        c->nets[i].b1h = m.h2[i].b1h; c->nets[i].b2h = m.h2[i].b2h;
        c->nets[i].h2_sig_descale = m.h2[i].sig_descale; c->nets[i].h2_out_descale = m.h2[i].out_descale;
    }
    return LCRC_OK;
}

// Fragment order of v_mfma_f32_16x16x4_f32's A operand (lane l: row l&15, k-slot l>>4).
//   w1p[(ht*nkq + kq)*64 + l][j] = W1[16ht + (l&15)][16kq + 4j + (l>>4)]
//   w2p[(ht*n_ot + ot)*64 + l][r] = W2[16ot + (l&15)][16ht + 4(l>>4) + r]
// Out-of-range rows/columns are zeros, which is what makes padded hidden units and
// padded k-steps contribute nothing (the reference zero-fills its x4 pads the same
// way, nn.cpp:239-243,276-280).
// Host half: shape + every array of the net in ONE block of floats (each array at a 64-float boundary), so that
// packing needs no GPU (it runs while the HIP runtime is still starting up, several nets in parallel) and the upload
// is one allocation and one copy per net.
struct PackedNet {
    NetDev d;                 // shape fields only
    std::vector<float> blob;
    size_t w1p, w2p, b1, b2, mean, dev;     // offsets (floats)
};

void pack_net_host(const HostNet &h, PackedNet &pk)
{
    NetDev &d = pk.d;
    memset(&d, 0, sizeof d);
    d.n_inp = h.n_inp; d.n_hid = h.n_hid; d.n_out = h.n_out;
    d.ksteps = (h.n_inp + 3) / 4;
    d.nkq = (d.ksteps + 3) / 4;
    d.nht = (h.n_hid + 15) / 16;
    d.n_ot = (h.n_out + 15) / 16;
    d.npairs = (d.nht + 1) / 2;
    d.h2_sig_descale = d.h2_out_descale = 1.f;
    // (+ one all-zero fragment behind each weight array: the run-time-shape kernels run the MFMA groups of their size
    //  class unconditionally and point the entries past this net's k-groups / output tiles at it)
    // (b1 padded to whole tile pairs, mean / dev to whole 32-deep k-steps: what the split-f16 kernels stage)
    const int ns = (h.n_inp + 31) / 32;
    const size_t n_w1 = ((size_t)d.nht * d.nkq + 1) * 256, n_w2 = ((size_t)d.nht * d.n_ot + 1) * 256;
    const size_t n_b1 = (size_t)d.npairs * 32, n_b2 = (size_t)d.n_ot * 16, n_nrm = (size_t)std::max(d.nkq * 16, ns * 32);
    auto up = [](size_t v) { return (v + 63) & ~(size_t)63; };
    pk.w1p = 0;
    pk.w2p = pk.w1p + up(n_w1);
    pk.b1 = pk.w2p + up(n_w2);
    pk.b2 = pk.b1 + up(n_b1);
    pk.mean = pk.b2 + up(n_b2);
    pk.dev = pk.mean + up(n_nrm);
    pk.blob.assign(pk.dev + up(n_nrm), 0.f);
    float *w1p = pk.blob.data() + pk.w1p, *w2p = pk.blob.data() + pk.w2p;
    for (int ht = 0; ht < d.nht; ht++)
        for (int kq = 0; kq < d.nkq; kq++)
            for (int l = 0; l < 64; l++) {
                const int hh = 16 * ht + (l & 15);
                if (hh >= h.n_hid) continue;
                const float *row = &h.w1[(size_t)hh * h.n_inp];
                float *dst = w1p + (((size_t)ht * d.nkq + kq) * 64 + l) * 4;
                for (int j = 0; j < 4; j++) {
                    const int k = 16 * kq + 4 * j + (l >> 4);
                    if (k < h.n_inp) dst[j] = row[k];
                }
            }
    for (int ht = 0; ht < d.nht; ht++)
        for (int ot = 0; ot < d.n_ot; ot++)
            for (int l = 0; l < 64; l++) {
                const int o = 16 * ot + (l & 15);
                if (o >= h.n_out) continue;
                const float *row = &h.w2[(size_t)o * h.n_hid];
                float *dst = w2p + (((size_t)ht * d.n_ot + ot) * 64 + l) * 4;
                for (int r = 0; r < 4; r++) {
                    const int hh = 16 * ht + 4 * (l >> 4) + r;
                    if (hh < h.n_hid) dst[r] = row[hh];
                }
            }
    memcpy(pk.blob.data() + pk.b1, h.b1.data(), sizeof(float) * h.n_hid);
    memcpy(pk.blob.data() + pk.b2, h.b2.data(), sizeof(float) * h.n_out);
    memcpy(pk.blob.data() + pk.mean, h.mean.data(), sizeof(float) * h.n_inp);
    std::fill(pk.blob.begin() + pk.dev, pk.blob.end(), 1.f);
    memcpy(pk.blob.data() + pk.dev, h.dev.data(), sizeof(float) * h.n_inp);
}

// Device half: the block to the GPU, pointers into it
int upload_net(lcrc_ctx *c, const PackedNet &pk, NetDev &d)
{
    const float *base = nullptr;
    HIP_TRY(c, dev_upload(c, pk.blob, &base));
    d = pk.d;
    d.w1p = reinterpret_cast<const float4 *>(base + pk.w1p);
    d.w2p = reinterpret_cast<const float4 *>(base + pk.w2p);
    d.b1 = base + pk.b1; d.b2 = base + pk.b2; d.mean = base + pk.mean; d.dev = base + pk.dev;
    d.w1h = d.w2h = nullptr;       // built on request (ensure_split_f16)
    d.b1h = d.b2h = nullptr;
    return LCRC_OK;
}

int pack_net(lcrc_ctx *c, const HostNet &h, NetDev &d)
{
    PackedNet pk;
    pack_net_host(h, pk);
    return upload_net(c, pk, d);
}

// Test hook (lcrc_debug_fail_alloc): the n-th buffer allocation from now on fails with out-of-memory.
std::atomic<int> g_fail_alloc{-1};
bool inject_alloc_failure()
{
    int v = g_fail_alloc.load();
    while (v >= 0) {
        if (g_fail_alloc.compare_exchange_weak(v, v - 1)) return v == 0;
    }
    return false;
}
// Test hook (lcrc_debug_fail_launch): the n-th posterior launch from now on fails before anything is queued.
std::atomic<int> g_fail_launch{-1};
bool inject_launch_failure()
{
    int v = g_fail_launch.load();
    while (v >= 0) {
        if (g_fail_launch.compare_exchange_weak(v, v - 1)) return v == 0;
    }
    return false;
}
hipError_t dev_alloc(void **p, size_t bytes)
{
    *p = nullptr;
    return inject_alloc_failure() ? hipErrorOutOfMemory : hipMalloc(p, bytes);
}
// Every pinned allocation is PORTABLE: a process that drives several GPUs (phnrec -g N: contexts on devices 0..N-1 in
// one address space) must be able to hand any of them to any device's copy engine, and a mapped buffer is read in
// place by the kernels of the context's OWN device, whose pointer is taken under that device (hipSetDevice precedes
// every hipHostGetDevicePointer here).  Without the flag the registration belongs to the device that was current at
// allocation time only -- invisible on a one-GPU box.
// device_reads: a buffer that kernels read in place (mapped into the device, coherent: never cached on the device side)
// Large buffers (the posterior, feature and byte buffers of a list's launches: 5-60 MB each) are the process's own
// 2 MiB-aligned anonymous memory with MADV_HUGEPAGE, touched, then registered with the runtime (hipHostRegister, mapped
// and portable): pinning and mapping 2 MiB pages instead of 4 KiB ones takes a third of hipHostMalloc's time -- 3 x 40 MiB:
// 10.7 against 32 ms; 6 x 40 MiB: 23 against 67 ms -- and the kernel tears such a process down 30-40 ms sooner after exit
// (tools/ubench/pin_probe, profiles/r06_ab_runs.txt 6).  Where transparent huge pages are off the advice is a no-op and
// the cost is hipHostMalloc's.  Small buffers, and any failure on the way, take hipHostMalloc.  pinned_free() knows which.
namespace {
struct PinnedRegion { void *map_base; size_t map_bytes; };
std::mutex g_pinned_mu;
std::unordered_map<void *, PinnedRegion> g_pinned;
constexpr size_t kHugePage = 2u << 20, kHugeMin = 4u << 20;
}  // namespace

hipError_t pinned_alloc(void **p, size_t bytes, bool device_reads)
{
    *p = nullptr;
    if (inject_alloc_failure()) return hipErrorOutOfMemory;
#ifndef LCRC_PINNED_PLAIN
    if (bytes >= kHugeMin) {
        const size_t len = (bytes + kHugePage - 1) & ~(kHugePage - 1), map_bytes = len + kHugePage;
        void *base = mmap(nullptr, map_bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        if (base != MAP_FAILED) {
            void *h = (void *)(((uintptr_t)base + kHugePage - 1) & ~(uintptr_t)(kHugePage - 1));
            (void)madvise(h, len, MADV_HUGEPAGE);
            for (size_t o = 0; o < len; o += 4096) ((volatile char *)h)[o] = 0;      // (the pages exist before they are pinned)
            if (hipHostRegister(h, len, hipHostRegisterMapped | hipHostRegisterPortable) == hipSuccess) {
                std::lock_guard<std::mutex> l(g_pinned_mu);
                g_pinned[h] = PinnedRegion{base, map_bytes};
                *p = h;
                return hipSuccess;
            }
            (void)hipGetLastError();
            (void)munmap(base, map_bytes);
        }
    }
#endif
    return hipHostMalloc(p, bytes, device_reads ? kPinnedMapped : kPinned);
}

hipError_t pinned_free(void *p)
{
    if (!p) return hipSuccess;
    PinnedRegion r{nullptr, 0};
    {
        std::lock_guard<std::mutex> l(g_pinned_mu);
        auto it = g_pinned.find(p);
        if (it != g_pinned.end()) { r = it->second; g_pinned.erase(it); }
    }
    if (!r.map_base) return hipHostFree(p);
    const hipError_t e = hipHostUnregister(p);
    (void)munmap(r.map_base, r.map_bytes);
    return e;
}

void free_frame_staging(lcrc_ctx *c)
{
    if (c->dec_stream) (void)hipStreamSynchronize(c->dec_stream);      // (a decoder of the last call may still read d_post)
    if (c->d_mel) (void)hipFree(c->d_mel);
    if (c->d_post) (void)hipFree(c->d_post);
    if (c->h_mel) (void)pinned_free(c->h_mel);
    if (c->h_post) (void)pinned_free(c->h_post);
    c->d_mel = c->d_post = c->h_mel = c->h_post = nullptr;
    c->cap_rows = c->cap_host_post = c->d_post_cap = 0;
}

void free_offset_staging(lcrc_ctx *c)
{
    if (c->d_off) (void)hipFree(c->d_off);
    if (c->h_off) (void)pinned_free(c->h_off);
    c->d_off = c->h_off = nullptr;
    c->cap_utts = 0;
}

// The end of a host-pointer call's work on the context's stream.  Default: hipStreamSynchronize, which spins on the
// completion signal (lowest latency; a core per waiting thread).  Polling mode (lcrc_set_wait_mode): an event behind the
// work, queried between short sleeps -- the waiting thread uses next to no CPU time and learns of the completion some tens
// of microseconds late.  For callers with more contexts in flight than cores to spare (the CLI with many GPUs).
hipError_t wait_stream(lcrc_ctx *c);
hipError_t wait_event(lcrc_ctx *c, hipEvent_t ev)
{
    if (c->poll_wait_us <= 0) return hipEventSynchronize(ev);
    const timespec nap = {0, (long)c->poll_wait_us * 1000L};
    for (;;) {
        hipError_t e = hipEventQuery(ev);
        if (e != hipErrorNotReady) return e;
        nanosleep(&nap, nullptr);
    }
}

// the caller learns that its posterior kernels are done (the rest of the call -- decoder, copies -- is queued behind them)
hipError_t report_kernel_done(lcrc_ctx *c)
{
    if (!c->kdone_armed) return hipSuccess;
    c->kdone_armed = false;
    hipError_t e = wait_event(c, c->ev_kdone);
    if (e == hipSuccess && c->kdone_fn) c->kdone_fn(c->kdone_arg);
    return e;
}

// Device -> caller's (pageable) buffer through the pinned staging buffer: the copy engine moves piece k + 1 while the
// host copies piece k out, instead of one DMA followed by one memcpy of the whole (4.5 MB of posteriors for 8192 CZ
// frames: 90 us + 80 us).  Pieces of >= 512 KiB, at most four (lcrc_posteriors of 8192 CZ frames, median of 60 calls on
// one box: 0.469 / 0.465 / 0.431 / 0.454 ms with 1 / 2 / 4 / 8 pieces: an event wait per piece costs, too); small
// transfers take the plain road.
hipError_t copy_back(lcrc_ctx *c, float *dst, float *pinned, const float *dev, size_t nbytes)
{
    // (the kernel-done report comes after the copies have been queued: the copy engine starts behind the kernel either way)
    int pieces = (int)std::min<size_t>(kCopyPieces, nbytes / (512u << 10));
    if (!dst || pieces < 2) {
        hipError_t e = hipMemcpyAsync(pinned, dev, nbytes, hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = wait_stream(c);
        if (e == hipSuccess && dst) memcpy(dst, pinned, nbytes);
        return e;
    }
    const size_t step = ((nbytes / pieces) + 255) & ~(size_t)255;
    for (int k = 0; k < pieces; k++) {
        if (!c->ev_piece[k]) {
            hipError_t e = hipEventCreateWithFlags(&c->ev_piece[k], hipEventDisableTiming);
            if (e != hipSuccess) return e;
        }
        const size_t lo = (size_t)k * step, n = k + 1 == pieces ? nbytes - lo : step;
        hipError_t e = hipMemcpyAsync((char *)pinned + lo, (const char *)dev + lo, n, hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipEventRecord(c->ev_piece[k], c->stream);
        if (e != hipSuccess) return e;
    }
    {
        hipError_t e = report_kernel_done(c);
        if (e != hipSuccess) return e;
    }
    for (int k = 0; k < pieces; k++) {
        hipError_t e = wait_event(c, c->ev_piece[k]);
        if (e != hipSuccess) return e;
        const size_t lo = (size_t)k * step, n = k + 1 == pieces ? nbytes - lo : step;
        memcpy((char *)dst + lo, (const char *)pinned + lo, n);
    }
    return hipSuccess;
}

hipError_t wait_stream(lcrc_ctx *c)
{
    {
        hipError_t e = report_kernel_done(c);    // (a no-op unless a launch of this call armed it)
        if (e != hipSuccess) return e;
    }
    if (c->poll_wait_us <= 0) return hipStreamSynchronize(c->stream);
    hipError_t e = hipSuccess;
    if (!c->ev_wait) e = hipEventCreateWithFlags(&c->ev_wait, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventRecord(c->ev_wait, c->stream);
    if (e != hipSuccess) return e;
    const timespec nap = {0, (long)c->poll_wait_us * 1000L};
    for (;;) {
        e = hipEventQuery(c->ev_wait);
        if (e != hipErrorNotReady) return e;
        nanosleep(&nap, nullptr);
    }
}

// Staging buffers of the host-pointer entry points.  A failed allocation leaves the group it belongs to
// EMPTY (nothing half-allocated, capacity 0): the call fails with LCRC_E_NOMEM and a later call starts over.
int ensure_staging(lcrc_ctx *c, size_t rows, size_t utts)
{
    if (rows > c->cap_rows) {
        const size_t cap = rows + rows / 4 + 64;
        free_frame_staging(c);
        const size_t O = c->nets[2].n_out;
        // (the pinned posterior buffer -- the largest: 744 B per HU frame, ~8 ms of page pinning per 32 768 frames --
        //  follows in ensure_host_post, when a call actually copies posteriors back: with the decoder on the device
        //  and no read-back nothing ever does)
        if (dev_alloc((void **)&c->d_mel, cap * c->nbanks * sizeof(float)) != hipSuccess ||
            dev_alloc((void **)&c->d_post, cap * O * sizeof(float)) != hipSuccess ||
            pinned_alloc((void **)&c->h_mel, cap * c->nbanks * sizeof(float), true) != hipSuccess) {
            free_frame_staging(c);
            (void)hipGetLastError();
            return fail(c, LCRC_E_NOMEM, "cannot allocate staging buffers for " + std::to_string(cap) + " frames");
        }
        c->cap_rows = c->d_post_cap = cap;
    }
    if (utts + 1 > c->cap_utts) {
        const size_t cap = utts + utts / 4 + 64;
        free_offset_staging(c);
        if (dev_alloc((void **)&c->d_off, cap * sizeof(int)) != hipSuccess ||
            pinned_alloc((void **)&c->h_off, cap * sizeof(int)) != hipSuccess) {
            free_offset_staging(c);
            (void)hipGetLastError();
            return fail(c, LCRC_E_NOMEM, "cannot allocate staging buffers for " + std::to_string(cap) + " utterances");
        }
        c->cap_utts = cap;
    }
    return LCRC_OK;
}

// The pinned posterior buffer for as many rows as the device buffers hold (after ensure_staging).
int ensure_host_post(lcrc_ctx *c)
{
    if (c->h_post && c->cap_host_post >= c->cap_rows) return LCRC_OK;
    if (c->h_post) (void)pinned_free(c->h_post);
    c->h_post = nullptr; c->cap_host_post = 0;
    const size_t O = c->nets[2].n_out;
    if (pinned_alloc((void **)&c->h_post, c->cap_rows * O * sizeof(float), true) != hipSuccess) {      // (mapped: direct output)
        c->h_post = nullptr;
        (void)hipGetLastError();
        return fail(c, LCRC_E_NOMEM, "cannot allocate the pinned posterior buffer for " + std::to_string(c->cap_rows) + " frames");
    }
    c->cap_host_post = c->cap_rows;
    return LCRC_OK;
}

// Direct output, for the entry points whose caller reads the posteriors where the context keeps them (lcrc_stage_run,
// lcrc_wave_stage_run with post == NULL): the kernels store them straight into the pinned host buffer, over PCIe while they
// run, instead of into device memory with a copy command behind the launch.  One synchronous 8192-frame lcrc_stage_run:
// 0.378 -> 0.314 ms (the copy was 82 us at PCIe's rate plus its hand-over); lists: the same rate within their noise
// (-F 31.3 -> 31.9 M frames/s), with 24 MB per launch less in the copy queue that all contexts of a device share.  Not for
// callers that get the posteriors copied into their own buffer: there the copy-back's pieces overlap that host copy
// (lcrc_posteriors 0.47 -> 0.50 ms with direct output).  *out = where the launch stores.
int output_target(lcrc_ctx *c, bool copy_post, bool in_place, float **out, bool *direct)
{
    *out = c->d_post;
    *direct = false;
    if (!in_place || !copy_post || c->dec_P > 0) return LCRC_OK;      // (the decoder on the device reads d_post)
    const int rc = ensure_host_post(c);
    if (rc) return rc;
    HIP_TRY(c, hipHostGetDevicePointer((void **)out, c->h_post, 0));
    *direct = true;
    return LCRC_OK;
}

void fill_output_transform(const lcrc_ctx *c, int *func, float (*oc)[4], float (*ol)[2], int *be)
{
    for (int i = 0; i < 2; i++) {
        const lcrc_softening &sf = c->soft[i];
        func[i] = sf.func;
        if (sf.func == LCRC_SOFT_IGOR) {         // SoftIgor's sub-expressions, srec.cpp:166-171
            oc[i][0] = sf.arg1;
            oc[i][1] = 1.0f / sf.arg1;
            oc[i][2] = 1.0f / (1.0f - sf.arg1);
            ol[i][0] = logf(sf.arg3);
            ol[i][1] = logf(sf.arg2);
        }
    }
    *be = c->out_be;
}

// 1BT_DCT: ONE launch (the merger's kernel computes its C0 / DCT input rows from the mel tile it stages).
// 1BT / 3BT: features -> band nets (one launch, grid.y = band) -> merger, on the same stream.
int launch_traps(lcrc_ctx *c, const float *d_mel, const int *d_off, int n_utts, int n_rows, float *d_post,
                 hipStream_t s)
{
    if (n_rows <= 0) return LCRC_OK;
    const size_t Km = c->nets[2].n_inp;
    const bool fused_dct = c->system == SYS_1BT_DCT && !c->traps_unfused;
    const bool fused_bt = c->system != SYS_1BT_DCT && !c->traps_unfused && !c->bt_unfused;
    if (!fused_dct && !fused_bt && (size_t)n_rows > c->cap_feat_rows) {
        const size_t cap = (size_t)n_rows + n_rows / 4 + 64;
        if (c->d_feat) (void)hipFree(c->d_feat);
        if (c->d_minp) (void)hipFree(c->d_minp);
        c->d_feat = c->d_minp = nullptr;
        c->cap_feat_rows = 0;
        const size_t feat = c->system == SYS_1BT_DCT ? cap * Km
                            : c->system == SYS_LCRC_GEN ? cap * 2 * (size_t)c->band_nets[0].n_inp
                                                        : cap * (size_t)c->trap_bands * c->trap_len;
        HIP_TRY(c, hipMalloc((void **)&c->d_feat, feat * sizeof(float)));
        if (c->system != SYS_1BT_DCT) HIP_TRY(c, hipMalloc((void **)&c->d_minp, cap * Km * sizeof(float)));
        c->cap_feat_rows = cap;
    }
    TrapsFeatParams f;
    memset(&f, 0, sizeof f);
    f.mel = d_mel; f.off = d_off; f.n_utts = n_utts; f.n_rows = n_rows;
    f.nbanks = c->nbanks; f.trap_bands = c->trap_bands;
    f.mode = c->system == SYS_1BT_DCT ? 1 : c->system == SYS_LCRC_GEN ? 2 : 0;
    f.use_hamming = c->use_hamming ? 1 : 0; f.add_c0 = c->add_c0 ? 1 : 0; f.shift = c->shift;
    f.hamming = c->d_hamm31; f.costab = c->d_costab31; f.normc = c->normc31;
    f.trap_len = c->trap_len; f.back = c->trap_len - 1 - (c->trap_len - 1) / 2; f.half = (c->trap_len - 1) / 2 + 1;
    f.win = c->d_win_gen;
    f.out = c->d_feat;
    if (c->timing) HIP_TRY(c, hipEventRecord(c->ev0, s));
    MlpParams m;
    memset(&m, 0, sizeof m);
    m.n_rows = n_rows;
    m.tile_frames = c->tile_frames;
    if (fused_bt) {
        // 1BT / 3BT in ONE launch: every wave runs whole band classifiers on the LDS-staged mel tile, the merger follows
        m.net = c->nets[2];
        m.nets_dev = c->d_band_nets; m.out_col = c->d_band_col; m.n_nets = c->trap_bands;
        m.lds_nkq = c->band_max.nkq; m.lds_n_ot = c->band_max.n_ot;
        m.dct = f;
        m.out = d_post; m.out_ld = c->nets[2].n_out;
        fill_output_transform(c, m.out_func, m.out_c, m.out_l, &m.out_be);
        const hipError_t e = traps_1bt_launch(m, s, &c->mlp_variant);
        if (e == hipSuccess) {
            if (c->timing) { HIP_TRY(c, hipEventRecord(c->ev1, s)); c->timed = true; }
            return LCRC_OK;
        }
        if (e != hipErrorNotSupported) return fail(c, LCRC_E_DEVICE, std::string("traps_1bt_launch: ") + hipGetErrorString(e));
        (void)hipGetLastError();
        c->bt_unfused = true;                // no size class holds this model: the three-launch form from now on
        return launch_traps(c, d_mel, d_off, n_utts, n_rows, d_post, s);
    }
    if (!fused_dct) HIP_TRY(c, traps_features_launch(f, s));
    const float *merger_in = c->d_feat;
    if (c->system != SYS_1BT_DCT) {
        if (c->system == SYS_LCRC_GEN) {
            // the two band nets one after the other (their inputs may be wider than the many-nets kernels' 256): each
            // writes ln(p) into its columns of the merger's input rows
            const long in_ld = c->band_nets[0].n_inp;
            int col = 0;
            for (int i = 0; i < 2; i++) {
                m.net = c->band_nets[i];
                m.in = c->d_feat + (size_t)i * n_rows * in_ld; m.in_ld = in_ld;
                m.out = c->d_minp + col; m.out_ld = (long)Km;
                m.neg_log = 2;
                HIP_TRY(c, mlp_launch(m, s));
                col += c->band_nets[i].n_out;
            }
        } else {
            m.net = c->band_max;                 // one launch, grid.y = band net
            m.nets_dev = c->d_band_nets; m.out_col = c->d_band_col; m.n_nets = (int)c->band_nets.size();
            m.in = c->d_feat; m.in_ld = c->trap_len; m.in_net_stride = (long)n_rows * c->trap_len;
            m.out = c->d_minp; m.out_ld = (long)Km;
            m.neg_log = 1;
            HIP_TRY(c, mlp_launch(m, s));
            m.nets_dev = nullptr; m.out_col = nullptr; m.n_nets = 0;
        }
        merger_in = c->d_minp;
    }
    m.net = c->nets[2];
    m.in = merger_in; m.in_ld = (long)Km;
    if (fused_dct) { m.dct = f; m.in = nullptr; }
    m.out = d_post; m.out_ld = c->nets[2].n_out;
    m.neg_log = 0;
    fill_output_transform(c, m.out_func, m.out_c, m.out_l, &m.out_be);
    HIP_TRY(c, mlp_launch(m, s, &c->mlp_variant));
    if (c->timing) { HIP_TRY(c, hipEventRecord(c->ev1, s)); c->timed = true; }
    return LCRC_OK;
}

// Scratch of the split-hidden path (small launches), allocated when the first launch that could use it arrives: zeroed
// once, the kernels leave the tickets at zero.  A context that only ever sees large launches (the CLI's batches, the
// bench) never pays for it.  On failure the context simply keeps to the fused kernel.
void ensure_split_scratch(lcrc_ctx *c)
{
    if (c->d_part || c->split_scratch_failed) return;
    size_t pb = 0, gb = 0, cb = 0;
    lcrc_split_scratch(c->nets, kSplitCapWgs, &pb, &gb, &cb);
    void *part = nullptr, *gimg = nullptr, *cnt = nullptr;
    if (hipMalloc(&part, pb) != hipSuccess || hipMalloc(&gimg, gb) != hipSuccess || hipMalloc(&cnt, cb) != hipSuccess ||
        hipMemsetAsync(gimg, 0, gb, c->stream) != hipSuccess || hipMemsetAsync(cnt, 0, cb, c->stream) != hipSuccess) {
        if (part) (void)hipFree(part);
        if (gimg) (void)hipFree(gimg);
        if (cnt) (void)hipFree(cnt);
        (void)hipGetLastError();
        c->split_scratch_failed = true;
        return;
    }
    c->allocs.push_back(part); c->allocs.push_back(gimg); c->allocs.push_back(cnt);
    c->d_part = static_cast<float4 *>(part); c->d_gimg = static_cast<float4 *>(gimg); c->d_cnt = static_cast<unsigned *>(cnt);
}

// Rows [row_first, row_first + row_count) of the n_rows rows are computed (row_count < 0: all of them);
// d_post receives row_first's posteriors first.
// lcrc_set_kernel_done_callback: an event behind the posterior kernels of the current call; reported by
// report_kernel_done() once everything else of the call has been queued
// lcrc_set_launch_order: one gate per device, shared by the process's contexts.  A launch's posterior kernels wait (on the
// device: hipStreamWaitEvent) for the posterior kernels of the launch queued before it through the gate, and leave an event
// for the next one.  The events come from a small ring the gate owns (a waiter refers to the record that was current when it
// was queued; sixteen launches later nobody still waits for it).
// The gate lives as long as ordered contexts of its device do (`users`): with the last one its events are destroyed and
// `last` forgotten -- a later ordered context never waits on an event recorded on a stream that is gone.
struct LaunchGate {
    std::mutex mu;
    hipEvent_t ring[16] = {};
    int next = 0;
    hipEvent_t last = nullptr;
    int users = 0;
};
LaunchGate g_gates[64];

// lcrc_set_launch_order / lcrc_destroy: the context joins or leaves its device's gate
void gate_membership(lcrc_ctx *c, bool ordered)
{
    if (c->launch_ordered == ordered) return;
    c->launch_ordered = ordered;
    if (c->device < 0 || c->device >= 64) return;
    LaunchGate &g = g_gates[c->device];
    std::lock_guard<std::mutex> l(g.mu);
    g.users += ordered ? 1 : -1;
    if (g.users > 0) return;
    g.users = 0;
    (void)hipSetDevice(c->device);
    for (hipEvent_t &e : g.ring) { if (e) (void)hipEventDestroy(e); e = nullptr; }
    g.last = nullptr;
    g.next = 0;
}

struct GateHold {
    LaunchGate *g = nullptr;
    hipStream_t s;
    GateHold(lcrc_ctx *c, hipStream_t stream) : s(stream)
    {
        // (only launches on the context's OWN stream: an asynchronous lcrc_posteriors_device launch on a caller's stream stays
        //  out -- the gate would otherwise keep an event of a stream whose life it knows nothing about)
        if (!c->launch_ordered || c->device < 0 || c->device >= 64 || stream != c->stream) return;
        g = &g_gates[c->device];
        g->mu.lock();
    }
    hipError_t wait_for_previous() { return g && g->last ? hipStreamWaitEvent(s, g->last, 0) : hipSuccess; }
    hipError_t leave_event()
    {
        if (!g) return hipSuccess;
        hipEvent_t &e = g->ring[g->next];
        if (!e) { const hipError_t rc = hipEventCreateWithFlags(&e, hipEventDisableTiming); if (rc != hipSuccess) return rc; }
        const hipError_t rc = hipEventRecord(e, s);
        if (rc == hipSuccess) { g->last = e; g->next = (g->next + 1) % 16; }
        return rc;
    }
    ~GateHold() { if (g) g->mu.unlock(); }
};

int arm_kernel_done(lcrc_ctx *c, hipStream_t s)
{
    if (!c->kdone_fn) return LCRC_OK;
    if (!c->ev_kdone) HIP_TRY(c, hipEventCreateWithFlags(&c->ev_kdone, hipEventDisableTiming));
    HIP_TRY(c, hipEventRecord(c->ev_kdone, s));
    c->kdone_armed = true;
    return LCRC_OK;
}

int launch(lcrc_ctx *c, const float *d_mel, const int *d_off, int n_utts, int n_rows, float *d_post,
           hipStream_t s, float *const *dbg, int row_first, int row_count, bool timed)
{
    if (row_count < 0) { row_first = 0; row_count = n_rows; }
    if (inject_launch_failure()) return fail(c, LCRC_E_DEVICE, "injected launch failure (lcrc_debug_fail_launch)");
    // (a call outside the overlapped path must not write posteriors a still-running decoder of this context reads)
    if (!c->overlapped_call) { const int rc = settle_pending_decoders(c); if (rc) return rc; }
    if (c->system != SYS_LCRC) {
        if (dbg) return fail(c, LCRC_E_UNSUPPORTED, "stage probes exist for posteriors/system=LCRC only");
        if (row_first != 0 || row_count != n_rows)
            return fail(c, LCRC_E_UNSUPPORTED, "row ranges exist for posteriors/system=LCRC only");
        GateHold gate(c, s);
        HIP_TRY(c, gate.wait_for_previous());
        int rc = launch_traps(c, d_mel, d_off, n_utts, n_rows, d_post, s);
        HIP_TRY(c, gate.leave_event());
        if (rc == LCRC_OK) rc = arm_kernel_done(c, s);
        return rc;
    }
    if (row_count == 0) return LCRC_OK;
    // (the scratch's clears run on the context's stream: a launch on another stream waits for them once)
    if (!c->d_part && c->split_hint != 1 && c->arith == 0 && !dbg) {      // (any size: a large launch may end in a split tail)
        ensure_split_scratch(c);
        if (c->d_part && s != c->stream) HIP_TRY(c, hipStreamSynchronize(c->stream));
    }
    LcrcParams p;
    memset(&p, 0, sizeof p);
    for (int i = 0; i < 3; i++) p.net[i] = c->nets[i];
    p.mel = d_mel; p.off = d_off; p.post = d_post;
    p.win = c->d_win; p.costab = c->d_costab; p.normc = c->normc;
    p.n_utts = n_utts; p.n_rows = n_rows; p.nbanks = c->nbanks;
    fill_output_transform(c, p.out_func, p.out_c, p.out_l, &p.out_be);
    p.tile_frames = c->tile_frames;
    p.arith = c->arith;
    p.stamps = c->d_stamps;
    p.row_first = row_first; p.row_end = row_first + row_count;
    p.part = c->d_part; p.gimg = c->d_gimg; p.cnt = c->d_cnt;
    p.split_cap_wgs = c->d_part ? kSplitCapWgs : 0;
    p.split_hint = c->split_hint;
    if (dbg) { p.dbg_in0 = dbg[0]; p.dbg_in1 = dbg[1]; p.dbg_p0 = dbg[2]; p.dbg_p1 = dbg[3]; p.dbg_g = dbg[4]; }
    timed = timed && c->timing;
    GateHold gate(c, s);
    HIP_TRY(c, gate.wait_for_previous());
    if (timed) HIP_TRY(c, hipEventRecord(c->ev0, s));
    HIP_TRY(c, lcrc_launch(p, s, nullptr));
    if (timed) { HIP_TRY(c, hipEventRecord(c->ev1, s)); c->timed = true; }
    HIP_TRY(c, gate.leave_event());
    return arm_kernel_done(c, s);
}

// label and count buffers of the decoder on the device (one label slot per frame, one count per utterance)
// Posteriors for a caller's own buffer, launches of 8192 frames and more (posteriors/system=LCRC): the rows are computed in
// TWO launches that store straight into the pinned buffer, and the host copies the first half into the caller's buffer while
// the second half is being computed.  What such a call waits for is the host's copy out of the pinned buffer (4.5 MB per 8192
// CZ frames: ~0.2 ms, as long as the kernel) -- behind ONE launch it can only overlap the copy engine's pieces, behind the
// first of two it overlaps a kernel.  Two half launches cost 0.204 instead of 0.192 ms of kernel time (one workgroup per CU
// each instead of pairs) and save ~0.1 ms of waiting.  Same bits (a row's posteriors do not depend on how a launch is cut).
// *done = false: not applicable, the caller takes the ordinary road.
int two_part_output(lcrc_ctx *c, const float *d_mel, const int *d_off, int n_utts, int n, float *post, bool *done)
{
    constexpr long kMinRows = 8192;
    *done = false;
    if (!post || c->system != SYS_LCRC || c->dec_P > 0 || n < kMinRows) return LCRC_OK;
    int rc = ensure_host_post(c);
    if (rc) return rc;
    float *out = nullptr;
    HIP_TRY(c, hipHostGetDevicePointer((void **)&out, c->h_post, 0));
    const size_t O = c->nets[2].n_out;
    const int half = (n / 2) & ~15;
    if (!c->ev_piece[0]) HIP_TRY(c, hipEventCreateWithFlags(&c->ev_piece[0], hipEventDisableTiming));
    if (c->timing) HIP_TRY(c, hipEventRecord(c->ev0, c->stream));
    rc = launch(c, d_mel, d_off, n_utts, n, out, c->stream, nullptr, 0, half, false);
    if (rc) return rc;
    HIP_TRY(c, hipEventRecord(c->ev_piece[0], c->stream));
    rc = launch(c, d_mel, d_off, n_utts, n, out + (size_t)half * O, c->stream, nullptr, half, n - half, false);
    if (rc) return rc;
    if (c->timing) { HIP_TRY(c, hipEventRecord(c->ev1, c->stream)); c->timed = true; }
    HIP_TRY(c, wait_event(c, c->ev_piece[0]));
    memcpy(post, c->h_post, (size_t)half * O * sizeof(float));
    HIP_TRY(c, wait_stream(c));
    memcpy(post + (size_t)half * O, c->h_post + (size_t)half * O, (size_t)(n - half) * O * sizeof(float));
    *done = true;
    return LCRC_OK;
}

int run_host(lcrc_ctx *c, const float *mel, const int *off, int n_utts, int n, float *post,
             float *const *probes, bool decode = true)
{
    const size_t nb = c->nbanks, O = c->nets[2].n_out;
    int rc = ensure_staging(c, n, n_utts);
    if (rc) return rc;
    memcpy(c->h_mel, mel, (size_t)n * nb * sizeof(float));
    HIP_TRY(c, hipMemcpyAsync(c->d_mel, c->h_mel, (size_t)n * nb * sizeof(float), hipMemcpyHostToDevice, c->stream));
    const int *d_off = nullptr;
    if (off) {
        memcpy(c->h_off, off, (size_t)(n_utts + 1) * sizeof(int));
        HIP_TRY(c, hipMemcpyAsync(c->d_off, c->h_off, (size_t)(n_utts + 1) * sizeof(int), hipMemcpyHostToDevice, c->stream));
        d_off = c->d_off;
    }
    float *dbg[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    const size_t K = c->nets[0].n_inp, Ob = c->nets[0].n_out;
    const size_t widths[5] = {K, K, Ob, Ob, 2 * Ob};
    bool any = false;
    if (probes) {
        for (int i = 0; i < 5; i++) any = any || probes[i];
        if (any) {
            if ((size_t)n > c->cap_dbg) {
                for (int i = 0; i < 5; i++) { if (c->d_dbg[i]) (void)hipFree(c->d_dbg[i]); c->d_dbg[i] = nullptr; }
                c->cap_dbg = 0;
                for (int i = 0; i < 5; i++) HIP_TRY(c, hipMalloc((void **)&c->d_dbg[i], (size_t)n * widths[i] * sizeof(float)));
                c->cap_dbg = n;
            }
            for (int i = 0; i < 5; i++) dbg[i] = probes[i] ? c->d_dbg[i] : nullptr;
        }
    }
    if (!any && !(decode && c->dec_P > 0)) {
        bool done = false;
        rc = two_part_output(c, c->d_mel, d_off, off ? n_utts : 1, n, post, &done);
        if (rc || done) return rc;
    }
    float *out_dev = c->d_post;
    bool direct = false;
    rc = output_target(c, !(decode && c->dec_P > 0) || c->readback, post == nullptr, &out_dev, &direct);
    if (rc) return rc;
    rc = launch(c, c->d_mel, d_off, off ? n_utts : 1, n, out_dev, c->stream, any ? dbg : nullptr);
    if (rc) return rc;
    const bool decoding = decode && c->dec_P > 0;
    if (decoding) {
        if (!off) {                              // one utterance: the decoder still wants [0, n]
            c->h_off[0] = 0; c->h_off[1] = n;
            HIP_TRY(c, hipMemcpyAsync(c->d_off, c->h_off, 2 * sizeof(int), hipMemcpyHostToDevice, c->stream));
        }
        rc = decode_after(c, c->d_off, c->h_off, off ? n_utts : 1, n, c->d_post, c->stream);
        if (rc) return rc;
    }
    const bool copy_post = c->readback || !decoding;
    if (copy_post && direct) {
        HIP_TRY(c, wait_stream(c));
        if (post) memcpy(post, c->h_post, (size_t)n * O * sizeof(float));
    } else if (copy_post) {
        rc = ensure_host_post(c);
        if (rc) return rc;
        HIP_TRY(c, copy_back(c, post, c->h_post, c->d_post, (size_t)n * O * sizeof(float)));
    } else {
        HIP_TRY(c, wait_stream(c));
    }
    if (any)
        for (int i = 0; i < 5; i++)
            if (probes[i]) HIP_TRY(c, hipMemcpy(probes[i], c->d_dbg[i], (size_t)n * widths[i] * sizeof(float), hipMemcpyDeviceToHost));
    return LCRC_OK;
}

// Files + consistency checks shared by lcrc_create and lcrc_model_info (no GPU needed).
// half: taps per half context (windows/band{0,1}.window hold that many values); *ncoef: inputs per band of the band nets.
int load_model(const char *model_dir, int nbanks, HostNet *nets, std::vector<float> *win, int half = kHalf, int *ncoef = nullptr)
{
    const std::string dir(model_dir);
    const char *names[3] = {"band0", "band1", "merger"};
    for (int i = 0; i < 3; i++) {
        const std::string w = dir + "/weights/" + names[i] + ".weights";
        const std::string n = dir + "/norms/" + names[i] + ".norms";
        NetStatus s = load_net(w, n, nets[i]);
        if (s != NET_OK)   // the reference prints this and exit(1)s, traps.cpp:141-145,162-166
            return fail(nullptr, s == NET_NOWEIGHTS || s == NET_NONORMS ? LCRC_E_IO : LCRC_E_MODEL,
                        "ERROR: Loading neural network: weights " + w + ", norms " + n + " (" + net_status_str(s) + ")");
    }
    for (int i = 0; i < 2; i++) {
        const std::string w = dir + "/windows/band" + std::to_string(i) + ".window";
        if (!load_window(w, half, win[i]))
            return fail(nullptr, LCRC_E_IO, "ERROR: Unable load window: " + w);
    }
    // (the reference derives the coefficients per band from net 0 and strides both nets' rows by their own sizes,
    //  traps.cpp:318-334: anything but equal sizes that nbanks divides reads or leaves garbage there)
    if (nets[0].n_inp != nets[1].n_inp || nets[0].n_inp % nbanks != 0)
        return fail(nullptr, LCRC_E_MODEL, "band classifier input size " + std::to_string(nets[0].n_inp) +
                    " is not nbanks x (coefficients per band), nbanks = " + std::to_string(nbanks));
    if (ncoef) *ncoef = nets[0].n_inp / nbanks;
    else if (nets[0].n_inp != nbanks * kNCoef)
        return fail(nullptr, LCRC_E_MODEL, "band classifier input size " + std::to_string(nets[0].n_inp) +
                    " != nbanks*11 = " + std::to_string(nbanks * kNCoef));
    if (nets[0].n_out + nets[1].n_out != nets[2].n_inp)
        return fail(nullptr, LCRC_E_MODEL, "merger input size does not equal the two band classifiers' outputs");
    return LCRC_OK;
}

void shape_of(const HostNet &h, NetDev &d)
{
    memset(&d, 0, sizeof d);
    d.n_inp = h.n_inp; d.n_hid = h.n_hid; d.n_out = h.n_out;
    d.ksteps = (h.n_inp + 3) / 4;
    d.nkq = (d.ksteps + 3) / 4;
    d.nht = (h.n_hid + 15) / 16;
    d.n_ot = (h.n_out + 15) / 16;
}

}  // namespace lcrc_impl

using namespace lcrc_impl;

extern "C" {

int lcrc_abi_version(void) { return LCRC_ABI_VERSION; }

int lcrc_model_info(const char *model_dir, int nbanks, int *dims9, char *kernel, size_t kernel_cap,
                    unsigned *lds_bytes)
{
    if (!model_dir || nbanks <= 0) return fail(nullptr, LCRC_E_ARG, "lcrc_model_info: bad argument");
    HostNet nets[3];
    std::vector<float> win[2];
    int ncoef = 0;
    int rc = load_model(model_dir, nbanks, nets, win, kHalf, &ncoef);
    if (rc) return rc;
    if (ncoef != kNCoef)
        return fail(nullptr, LCRC_E_UNSUPPORTED, "band classifiers take " + std::to_string(ncoef) +
                    " inputs per band: a geometry of the general kernels, not of the fused kernel (11)");
    NetDev nd[3];
    for (int i = 0; i < 3; i++) {
        shape_of(nets[i], nd[i]);
        if (dims9) { dims9[3 * i] = nets[i].n_inp; dims9[3 * i + 1] = nets[i].n_hid; dims9[3 * i + 2] = nets[i].n_out; }
    }
    unsigned lds = 0;
    const char *v = lcrc_variant_for(nd, nbanks, &lds);
    if (lds_bytes) *lds_bytes = lds;
    if (kernel && kernel_cap) snprintf(kernel, kernel_cap, "%s", v ? v : "");
    if (!v) return fail(nullptr, LCRC_E_UNSUPPORTED, "model geometry not supported by the fused kernel");
    return LCRC_OK;
}

const char *lcrc_last_error(const lcrc_ctx *ctx) { return ctx ? ctx->err.c_str() : g_create_err.c_str(); }

int lcrc_model_outputs(const char *model_dir, const char *system)
{
    if (!model_dir || !system) return fail(nullptr, LCRC_E_ARG, "lcrc_model_outputs: NULL argument");
    if (strcmp(system, "LCRC") && strcmp(system, "1BT_DCT") && strcmp(system, "1BT") && strcmp(system, "3BT"))
        return fail(nullptr, LCRC_E_ARG, std::string("Unknown posterior estimator system: ") + system);
    const std::string dir(model_dir);
    const std::string w = dir + "/weights/merger.weights", n = dir + "/norms/merger.norms";
    HostNet merger;
    NetStatus s = load_net(w, n, merger);
    if (s != NET_OK)
        return fail(nullptr, s == NET_NOWEIGHTS || s == NET_NONORMS ? LCRC_E_IO : LCRC_E_MODEL,
                    "ERROR: Loading neural network: weights " + w + ", norms " + n + " (" + net_status_str(s) + ")");
    return merger.n_out;
}

// Device, stream and events of a fresh context (shared by every system)
static int open_context(lcrc_ctx **out, int nbanks, int device_id)
{
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(nullptr, LCRC_E_DEVICE, "no HIP device available (this library has no CPU path)");
    if (device_id < 0 || device_id >= ndev)
        return fail(nullptr, LCRC_E_DEVICE, "device_id out of range");
    HIP_TRY(nullptr, hipSetDevice(device_id));
    lcrc_ctx *c = new lcrc_ctx;
    c->device = device_id;
    c->nbanks = nbanks;
    c->model = std::make_shared<SharedModel>();
    c->model->device = device_id;
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreate(&c->ev0) != hipSuccess || hipEventCreate(&c->ev1) != hipSuccess) {
        g_create_err = "cannot create HIP stream/events";
        lcrc_destroy(c);
        return LCRC_E_DEVICE;
    }
    *out = c;
    return LCRC_OK;
}

// The systems that are composed of general kernels: 1BT_DCT / 1BT / 3BT at any posteriors/length (with fused forms at
// the usual 31), and LCRC at a geometry other than the shipped models' (SYS_LCRC_GEN).
static int create_traps(lcrc_ctx **out, const char *model_dir, int sys, int nbanks, int trap_len, int add_c0, int hamming, int device_id)
{
    *out = nullptr;
    if (nbanks <= (sys == SYS_3BT ? 2 : 0)) return fail(nullptr, LCRC_E_ARG, "lcrc_create_system: nbanks too small");
    if (trap_len < 2 || trap_len > kMaxTrapLen)
        return fail(nullptr, LCRC_E_ARG, "posteriors/length must lie in 2.." + std::to_string(kMaxTrapLen));
    const int trap_bands = sys == SYS_3BT ? nbanks - 2 : nbanks;        // traps.cpp:95-97
    const int half = (trap_len - 1) / 2 + 1;                            // traps.cpp:93,288
    const bool lcrc = sys == SYS_LCRC_GEN;

    // -- files first (traps.cpp:119-166)
    const std::string dir(model_dir);
    std::vector<HostNet> band(sys == SYS_1BT_DCT ? 0 : lcrc ? 2 : trap_bands);
    std::vector<float> win[2];
    HostNet merger;
    auto load = [&](const std::string &name, HostNet &net) -> int {
        const std::string w = dir + "/weights/" + name + ".weights", n = dir + "/norms/" + name + ".norms";
        NetStatus s = load_net(w, n, net);
        if (s != NET_OK)
            return fail(nullptr, s == NET_NOWEIGHTS || s == NET_NONORMS ? LCRC_E_IO : LCRC_E_MODEL,
                        "ERROR: Loading neural network: weights " + w + ", norms " + n + " (" + net_status_str(s) + ")");
        return LCRC_OK;
    };
    size_t merger_in = 0;
    for (int i = 0; i < (int)band.size(); i++) {
        int rc = load("band" + std::to_string(i), band[i]);
        if (rc) return rc;
        // the reference hands every band net exactly trap_len values per frame (traps.cpp:253-259)
        if (!lcrc && band[i].n_inp != trap_len)
            return fail(nullptr, LCRC_E_MODEL, "band classifier " + std::to_string(i) + " takes " +
                        std::to_string(band[i].n_inp) + " inputs, the trajectory has " + std::to_string(trap_len));
        merger_in += band[i].n_out;
    }
    if (lcrc) {
        for (int i = 0; i < 2; i++) {
            const std::string w = dir + "/windows/band" + std::to_string(i) + ".window";
            if (!load_window(w, half, win[i])) return fail(nullptr, LCRC_E_IO, "ERROR: Unable load window: " + w);
        }
        // (coefficients per band from net 0, both nets' rows strided by their own sizes, traps.cpp:318-334)
        if (band[0].n_inp != band[1].n_inp || band[0].n_inp % nbanks != 0 || band[0].n_inp / nbanks - (add_c0 ? 1 : 0) < 0)
            return fail(nullptr, LCRC_E_MODEL, "band classifier input size " + std::to_string(band[0].n_inp) +
                        " is not nbanks x (coefficients per band), nbanks = " + std::to_string(nbanks));
    }
    {
        int rc = load("merger", merger);
        if (rc) return rc;
    }
    // values per band of the C0 / DCT features: merger_input_shift (traps.cpp:170), LCRC: the band nets' inputs per band
    const int shift = lcrc ? band[0].n_inp / nbanks : merger.n_inp / trap_bands;
    if (sys == SYS_1BT_DCT) {
        if (shift * trap_bands != merger.n_inp || shift < 1 || shift - (add_c0 ? 1 : 0) > trap_len)
            return fail(nullptr, LCRC_E_MODEL, "merger input size " + std::to_string(merger.n_inp) +
                        " is not nbanks x (coefficients per band)");
    } else if ((int)merger_in != merger.n_inp) {
        return fail(nullptr, LCRC_E_MODEL, "merger input size does not equal the band classifiers' outputs");
    }

    lcrc_ctx *c = nullptr;
    {
        int rc = open_context(&c, nbanks, device_id);
        if (rc) return rc;
    }
    auto bail = [&](int code) { g_create_err = c->err; lcrc_destroy(c); return code; };
    c->system = sys;
    c->trap_len = trap_len;
    {
        const char *e = getenv("PHNREC_TRAPS_UNFUSED");
        c->traps_unfused = (e && *e == '1') || trap_len != kTrapLen || lcrc;     // the fused forms are written for 31
    }
    c->trap_bands = trap_bands;
    c->shift = shift;
    c->use_hamming = hamming != 0;
    c->add_c0 = add_c0 != 0;
    c->band_nets.resize(band.size());
    for (size_t i = 0; i < band.size(); i++) {
        int rc = pack_net(c, band[i], c->band_nets[i]);
        if (rc) return bail(rc);
        if (!mlp_supports(c->band_nets[i])) { c->err = "band classifier too large (<= 208 outputs)"; return bail(LCRC_E_UNSUPPORTED); }
    }
    if (!c->band_nets.empty()) {
        std::vector<int> col(c->band_nets.size());
        int acc_col = 0;
        c->band_max = c->band_nets[0];
        for (size_t i = 0; i < c->band_nets.size(); i++) {
            col[i] = acc_col;
            acc_col += c->band_nets[i].n_out;
            c->band_max.ksteps = std::max(c->band_max.ksteps, c->band_nets[i].ksteps);
            c->band_max.nkq = std::max(c->band_max.nkq, c->band_nets[i].nkq);
            c->band_max.n_ot = std::max(c->band_max.n_ot, c->band_nets[i].n_ot);
        }
        if (dev_upload(c, c->band_nets, &c->d_band_nets) != hipSuccess || dev_upload(c, col, &c->d_band_col) != hipSuccess) {
            c->err = "upload failed";
            return bail(LCRC_E_DEVICE);
        }
    }
    c->model->host[2] = merger;
    {
        int rc = pack_net(c, merger, c->nets[2]);
        if (rc) return bail(rc);
    }
    if (!mlp_supports(c->nets[2])) { c->err = "merger too large (<= 1024 inputs, <= 208 outputs)"; return bail(LCRC_E_UNSUPPORTED); }
    // Hamming window over ones (traps.cpp:107-109, dspc.h:162-167) and sDCT's basis (dspc.h:206-221) over n points:
    // the trajectory's length, LCRC: a half context's
    const int n = lcrc ? half : trap_len, n_basis = lcrc ? std::max(1, shift) : trap_len;
    std::vector<float> hamm(trap_len), cosv((size_t)n_basis * n, 0.f);
    for (int i = 0; i < trap_len; i++)
        hamm[i] = 1.0f * (0.54f - 0.46f * cosf(2.0f * (float)M_PI * i / (trap_len - 1)));
    const float pibyn = (float)M_PI / (float)n;
    for (int k = 0; k < n_basis; k++) {
        const float v = pibyn * (float)(k + 1);
        for (int j = 0; j < n; j++) cosv[(size_t)k * n + j] = cosf(v * ((float)j + 0.5f));
    }
    c->normc31 = sqrtf(2.0f / (float)n);
    const float *p = nullptr;
    if (lcrc) {
        std::vector<float> w2(win[0]);
        w2.insert(w2.end(), win[1].begin(), win[1].end());
        if (dev_upload(c, w2, &p) != hipSuccess) { c->err = "upload failed"; return bail(LCRC_E_DEVICE); }
        c->d_win_gen = const_cast<float *>(p);
    }
    if (dev_upload(c, hamm, &p) != hipSuccess) { c->err = "upload failed"; return bail(LCRC_E_DEVICE); }
    c->d_hamm31 = const_cast<float *>(p);
    if (dev_upload(c, cosv, &p) != hipSuccess) { c->err = "upload failed"; return bail(LCRC_E_DEVICE); }
    c->d_costab31 = const_cast<float *>(p);
    c->variant = sys == SYS_1BT_DCT ? "traps_1bt_dct" : sys == SYS_1BT ? "traps_1bt" : sys == SYS_3BT ? "traps_3bt" : "lcrc_general";
    *out = c;
    return LCRC_OK;
}


int lcrc_device_warmup(int device_id)
{
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(nullptr, LCRC_E_DEVICE, "no HIP device available (this library has no CPU path)");
    if (device_id < 0 || device_id >= ndev) return fail(nullptr, LCRC_E_DEVICE, "device_id out of range");
    HIP_TRY(nullptr, hipSetDevice(device_id));
    HIP_TRY(nullptr, hipFree(nullptr));                 // brings the device's primary context up
    // the posterior kernels' code object, while the caller's own thread creates its context (stream, weights)
    (void)lcrc_preload_code();
    return LCRC_OK;
}

int lcrc_device_preload(int device_id, int what)
{
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(nullptr, LCRC_E_DEVICE, "no HIP device available (this library has no CPU path)");
    if (device_id < 0 || device_id >= ndev) return fail(nullptr, LCRC_E_DEVICE, "device_id out of range");
    HIP_TRY(nullptr, hipSetDevice(device_id));
    if (what & LCRC_PRELOAD_FRONTEND) (void)frontend_preload_code();
    if (what & LCRC_PRELOAD_DECODER) (void)phndec_preload_code();
    (void)hipGetLastError();
    return LCRC_OK;
}

int lcrc_device_pci_bus_id(int device_id, char *buf, int len)
{
    if (!buf || len < 13) return fail(nullptr, LCRC_E_ARG, "lcrc_device_pci_bus_id: buffer of at least 13 bytes needed");
    buf[0] = 0;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(nullptr, LCRC_E_DEVICE, "no HIP device available (this library has no CPU path)");
    if (device_id < 0 || device_id >= ndev) return fail(nullptr, LCRC_E_DEVICE, "device_id out of range");
    HIP_TRY(nullptr, hipDeviceGetPCIBusId(buf, len, device_id));
    return LCRC_OK;
}

int lcrc_create(lcrc_ctx **out, const char *model_dir, int nbanks, int trap_len, int add_c0, int device_id)
{
    if (!out || !model_dir) return fail(nullptr, LCRC_E_ARG, "lcrc_create: NULL argument");
    *out = nullptr;
    if (nbanks <= 0) return fail(nullptr, LCRC_E_ARG, "lcrc_create: nbanks must be positive");
    if (trap_len < 2 || trap_len > kMaxTrapLen)
        return fail(nullptr, LCRC_E_ARG, "lcrc_create: posteriors/length must lie in 2.." + std::to_string(kMaxTrapLen));
    // Any geometry but the shipped models' (31 frames, C0 + 10 coefficients per band and half context) takes the general
    // kernels: correct, composed of three launches, not the fused kernel this library is about (lcrc.h).
    if (trap_len != kTrapLen || !add_c0) return create_traps(out, model_dir, SYS_LCRC_GEN, nbanks, trap_len, add_c0, 0, device_id);

    // -- files first, so that a bad model directory is reported even without a GPU
    StartupTrace trace;
    HostNet nets[3];
    std::vector<float> win[2];
    {
        int ncoef = 0;
        int rc = load_model(model_dir, nbanks, nets, win, kHalf, &ncoef);
        if (rc) return rc;
        if (ncoef != kNCoef) return create_traps(out, model_dir, SYS_LCRC_GEN, nbanks, trap_len, add_c0, 0, device_id);
    }
    trace.mark("model files");
    // fragment order on the host first (no GPU involved: a caller that started lcrc_device_warmup on another thread has
    // the HIP runtime coming up meanwhile), the three nets side by side
    PackedNet packed[3];
    {
        std::thread t1([&] { pack_net_host(nets[1], packed[1]); }), t2([&] { pack_net_host(nets[2], packed[2]); });
        pack_net_host(nets[0], packed[0]);
        t1.join();
        t2.join();
    }
    trace.mark("pack (host)");
    lcrc_ctx *c = nullptr;
    {
        int rc = open_context(&c, nbanks, device_id);
        if (rc) return rc;
    }
    trace.mark("HIP device, stream, events");
    auto bail = [&](int code) { g_create_err = c->err; lcrc_destroy(c); return code; };
    for (int i = 0; i < 3; i++) {
        int rc = upload_net(c, packed[i], c->nets[i]);
        if (rc) return bail(rc);
        c->model->host[i] = std::move(nets[i]);
    }
    trace.mark("upload weights");
    // DCT basis exactly as sDCT evaluates it (dspc.h:206-221), in f32 with libm cosf
    std::vector<float> cosv(10 * 16), winv(32);
    const float pibyn = (float)M_PI / (float)kHalf;
    for (int k = 0; k < 10; k++) {
        const float v = pibyn * (float)(k + 1);
        for (int j = 0; j < kHalf; j++) cosv[k * 16 + j] = cosf(v * ((float)j + 0.5f));
    }
    c->normc = sqrtf(2.0f / (float)kHalf);
    for (int i = 0; i < 2; i++) memcpy(&winv[i * 16], win[i].data(), 16 * sizeof(float));
    const float *p = nullptr;
    if (dev_upload(c, cosv, &p) != hipSuccess) { c->err = "upload failed"; return bail(LCRC_E_DEVICE); }
    c->d_costab = const_cast<float *>(p);
    if (dev_upload(c, winv, &p) != hipSuccess) { c->err = "upload failed"; return bail(LCRC_E_DEVICE); }
    c->d_win = const_cast<float *>(p);
    const char *v = lcrc_variant_for(c->nets, nbanks, &c->lds_bytes);
    if (!v) {
        c->err = "model geometry not supported by the fused kernel (needs <= 23 banks, <= 208 outputs, LDS " +
                 std::to_string(c->lds_bytes) + " B <= 160 KiB)";
        return bail(LCRC_E_UNSUPPORTED);
    }
    c->variant = v;
    c->trap_bands = 2;
    *out = c;
    return LCRC_OK;
}

int lcrc_create_system(lcrc_ctx **out, const char *model_dir, const char *system, int nbanks, int trap_len,
                       int add_c0, int hamming, int device_id)
{
    if (!out || !model_dir || !system) return fail(nullptr, LCRC_E_ARG, "lcrc_create_system: NULL argument");
    // LCRC ignores posteriors/hamming: its half-context windows come from files (traps.cpp:224-243 is skipped)
    if (!strcmp(system, "LCRC")) return lcrc_create(out, model_dir, nbanks, trap_len, add_c0, device_id);
    *out = nullptr;
    int sys;
    if (!strcmp(system, "1BT_DCT")) sys = SYS_1BT_DCT;
    else if (!strcmp(system, "1BT")) sys = SYS_1BT;
    else if (!strcmp(system, "3BT")) sys = SYS_3BT;
    else return fail(nullptr, LCRC_E_ARG, std::string("Unknown posterior estimator system: ") + system);   // srec.cpp:605-611
    return create_traps(out, model_dir, sys, nbanks, trap_len, add_c0, hamming, device_id);
}

int lcrc_clone(lcrc_ctx **out, const lcrc_ctx *src)
{
    if (!out || !src) return fail(nullptr, LCRC_E_ARG, "lcrc_clone: NULL argument");
    *out = nullptr;
    lcrc_ctx *c = nullptr;
    {
        int rc = open_context(&c, src->nbanks, src->device);
        if (rc) return rc;
    }
    c->model = src->model;                 // the read-only device buffers and the host nets are shared
    for (int i = 0; i < 3; i++) {
        c->nets[i] = src->nets[i];
        c->nets[i].w1h = c->nets[i].w2h = nullptr;      // (handed out by lcrc_set_arithmetic)
        c->nets[i].b1h = c->nets[i].b2h = nullptr;
    }
    c->d_win = src->d_win; c->d_costab = src->d_costab; c->normc = src->normc;
    c->system = src->system; c->trap_bands = src->trap_bands; c->shift = src->shift;
    c->use_hamming = src->use_hamming; c->add_c0 = src->add_c0;
    c->band_nets = src->band_nets; c->d_band_nets = src->d_band_nets; c->d_band_col = src->d_band_col;
    c->band_max = src->band_max;
    c->d_hamm31 = src->d_hamm31; c->d_costab31 = src->d_costab31; c->normc31 = src->normc31;
    c->trap_len = src->trap_len; c->d_win_gen = src->d_win_gen;
    c->traps_unfused = src->traps_unfused; c->bt_unfused = src->bt_unfused;
    c->variant = src->variant; c->lds_bytes = src->lds_bytes;
    *out = c;
    return LCRC_OK;
}

void lcrc_destroy(lcrc_ctx *c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->dec_stream) (void)hipStreamSynchronize(c->dec_stream);
    gate_membership(c, false);             // (behind the synchronise: nothing of this context is queued any more)
    for (void *p : c->allocs) (void)hipFree(p);
    free_frame_staging(c);
    if (c->alt.d_post) (void)hipFree(c->alt.d_post);
    if (c->alt.h_labels) (void)pinned_free(c->alt.h_labels);
    if (c->alt.h_count) (void)pinned_free(c->alt.h_count);
    if (c->d_dec_off) (void)hipFree(c->d_dec_off);
    if (c->alt.d_dec_off) (void)hipFree(c->alt.d_dec_off);
    if (c->ev_post) (void)hipEventDestroy(c->ev_post);
    if (c->ev_dec_done) (void)hipEventDestroy(c->ev_dec_done);
    if (c->alt.ev_dec_done) (void)hipEventDestroy(c->alt.ev_dec_done);
    if (c->dec_stream) (void)hipStreamDestroy(c->dec_stream);
    free_offset_staging(c);
    for (float *p : c->d_dbg) if (p) (void)hipFree(p);
    if (c->h_labels) (void)pinned_free(c->h_labels);
    if (c->h_count) (void)pinned_free(c->h_count);
    if (c->d_feat) (void)hipFree(c->d_feat);
    if (c->d_minp) (void)hipFree(c->d_minp);
    if (c->d_hamming) (void)hipFree(c->d_hamming);
    if (c->d_coeffs) (void)hipFree(c->d_coeffs);
    if (c->d_twiddle) (void)hipFree(c->d_twiddle);
    if (c->d_runs) (void)hipFree(c->d_runs);
    if (c->d_bytes) (void)hipFree(c->d_bytes);
    if (c->h_bytes) (void)pinned_free(c->h_bytes);
    if (c->d_soff) (void)hipFree(c->d_soff);
    if (c->h_soff) (void)pinned_free(c->h_soff);
    if (c->d_foff) (void)hipFree(c->d_foff);
    if (c->h_foff) (void)pinned_free(c->h_foff);
    if (c->d_means) (void)hipFree(c->d_means);
    if (c->d_mean_part) (void)hipFree(c->d_mean_part);
    if (c->h_ring) (void)pinned_free(c->h_ring);
    if (c->h_pushout) (void)pinned_free(c->h_pushout);
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    if (c->ev_wait) (void)hipEventDestroy(c->ev_wait);
    for (hipEvent_t e : c->ev_piece) if (e) (void)hipEventDestroy(e);
    if (c->ev_kdone) (void)hipEventDestroy(c->ev_kdone);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

int lcrc_num_outputs(const lcrc_ctx *c) { return c ? c->nets[2].n_out : LCRC_E_ARG; }
int lcrc_num_banks(const lcrc_ctx *c) { return c ? c->nbanks : LCRC_E_ARG; }
int lcrc_trap_shift(const lcrc_ctx *c) { return c ? (c->trap_len - 1) / 2 : LCRC_E_ARG; }
int lcrc_device(const lcrc_ctx *c) { return c ? c->device : LCRC_E_ARG; }
const char *lcrc_kernel_name(const lcrc_ctx *c) { return c ? c->variant : "none"; }

int lcrc_net_dims(const lcrc_ctx *c, int which, int *n_inp, int *n_hid, int *n_out)
{
    // band classifiers first (2 for LCRC, trap_bands for 1BT / 3BT, none for 1BT_DCT), then the merger
    if (!c) return LCRC_E_ARG;
    const int n_band = c->system == SYS_LCRC ? 2 : (int)c->band_nets.size();
    if (which < 0 || which > n_band) return LCRC_E_ARG;
    const NetDev &nd = which == n_band ? c->nets[2] : (c->system == SYS_LCRC ? c->nets[which] : c->band_nets[which]);
    if (n_inp) *n_inp = nd.n_inp;
    if (n_hid) *n_hid = nd.n_hid;
    if (n_out) *n_out = nd.n_out;
    return LCRC_OK;
}

int lcrc_posteriors(lcrc_ctx *c, const float *mel, int n, float *post)
{
    if (!c) return LCRC_E_ARG;
    if (n < 0 || (n > 0 && (!mel || !post))) return fail(c, LCRC_E_ARG, "lcrc_posteriors: bad argument");
    if (n == 0) return LCRC_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    return run_host(c, mel, nullptr, 1, n, post, nullptr);
}

int lcrc_posteriors_probe(lcrc_ctx *c, const float *mel, int n, float *post, float *in0, float *in1,
                          float *p0, float *p1, float *g)
{
    if (!c) return LCRC_E_ARG;
    if (n < 0 || (n > 0 && (!mel || !post))) return fail(c, LCRC_E_ARG, "lcrc_posteriors_probe: bad argument");
    if (c->system != SYS_LCRC) return fail(c, LCRC_E_UNSUPPORTED, "stage probes exist for posteriors/system=LCRC only");
    if (n == 0) return LCRC_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    float *probes[5] = {in0, in1, p0, p1, g};
    return run_host(c, mel, nullptr, 1, n, post, probes);
}

int lcrc_posteriors_batch(lcrc_ctx *c, const float *mel, const int *off, int n_utts, float *post)
{
    if (!c) return LCRC_E_ARG;
    if (n_utts < 0 || (n_utts > 0 && !off)) return fail(c, LCRC_E_ARG, "lcrc_posteriors_batch: bad argument");
    if (n_utts == 0) return LCRC_OK;
    if (off[0] != 0) return fail(c, LCRC_E_ARG, "lcrc_posteriors_batch: off[0] must be 0");
    for (int u = 0; u < n_utts; u++)
        if (off[u + 1] < off[u]) return fail(c, LCRC_E_ARG, "lcrc_posteriors_batch: offsets must be non-decreasing");
    const int n = off[n_utts];
    if (n == 0) return LCRC_OK;
    if (!mel || !post) return fail(c, LCRC_E_ARG, "lcrc_posteriors_batch: NULL buffer");
    HIP_TRY(c, hipSetDevice(c->device));
    return run_host(c, mel, off, n_utts, n, post, nullptr);
}

int lcrc_stage_buffers(lcrc_ctx *c, int rows, float **mel, float **post)
{
    if (!c || rows < 0 || !mel || !post) return LCRC_E_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    int rc = ensure_staging(c, rows > 0 ? rows : 1, 1);
    if (rc) return rc;
    rc = ensure_host_post(c);
    if (rc) return rc;
    *mel = c->h_mel;
    *post = c->h_post;
    return LCRC_OK;
}

int lcrc_stage_run(lcrc_ctx *c, const int *off, int n_utts)
{
    if (!c || !off || n_utts < 1 || off[0] != 0) return fail(c, LCRC_E_ARG, "lcrc_stage_run: bad argument");
    for (int u = 0; u < n_utts; u++)
        if (off[u + 1] < off[u]) return fail(c, LCRC_E_ARG, "lcrc_stage_run: offsets must be non-decreasing");
    const int n = off[n_utts];
    HIP_TRY(c, hipSetDevice(c->device));
    OverlapScope scope(c);
    { const int rc0 = begin_overlapped_call(c); if (rc0) return rc0; }       // (every staged call, empty ones too: the sets alternate per call)
    if (n == 0) { c->label_utts = 0; return LCRC_OK; }
    if ((size_t)n > c->cap_rows) return fail(c, LCRC_E_ARG, "lcrc_stage_run: more rows than lcrc_stage_buffers reserved");
    const size_t nb = c->nbanks, O = c->nets[2].n_out;
    // only the offsets still have to be staged; the frame buffers are the pinned ones already
    float *keep_mel = c->h_mel, *keep_post = c->h_post, *dm = c->d_mel, *dp = c->d_post;
    const size_t keep_cap = c->cap_rows;
    int rc = ensure_staging(c, keep_cap, n_utts);
    if (rc) return rc;
    if (c->h_mel != keep_mel || c->h_post != keep_post || c->d_mel != dm || c->d_post != dp)
        return fail(c, LCRC_E_NOMEM, "lcrc_stage_run: staging buffers moved");
    memcpy(c->h_off, off, (size_t)(n_utts + 1) * sizeof(int));
    HIP_TRY(c, hipMemcpyAsync(c->d_off, c->h_off, (size_t)(n_utts + 1) * sizeof(int), hipMemcpyHostToDevice, c->stream));
    // The kernel reads the features where they lie, in the pinned buffer (60-92 B per frame over PCIe, each row by two
    // workgroups): no copy command.  A copy would be cheap by itself, but copies of all contexts share the device's copy
    // queue, in order: this context's 2 MB would wait behind another context's 24 MB of posteriors on their way back,
    // which in turn wait for that context's kernel -- with three contexts in flight the launches ran strictly one after
    // the other, kernel / copy-back / kernel / ... (profiles/r03_ab_runs.txt 15: 20 -> 24-27 M frames/s for the CLI).
    // (the fused LCRC kernels stage a workgroup's rows once; the composed kernels of the other systems re-read theirs and
    //  get the copy)
    float *mel_in = c->d_mel;
    if (c->system == SYS_LCRC) HIP_TRY(c, hipHostGetDevicePointer((void **)&mel_in, c->h_mel, 0));
    else HIP_TRY(c, hipMemcpyAsync(c->d_mel, c->h_mel, (size_t)n * nb * sizeof(float), hipMemcpyHostToDevice, c->stream));
    float *out_dev = c->d_post;
    bool direct = false;
    rc = output_target(c, c->readback || c->dec_P <= 0, true, &out_dev, &direct);
    if (rc) return rc;
    rc = launch(c, mel_in, c->d_off, n_utts, n, out_dev, c->stream, nullptr);
    if (rc) return rc;
    rc = decode_after(c, c->d_off, c->h_off, n_utts, n, c->d_post, c->stream, true);
    if (rc) return rc;
    if (!direct && (c->readback || c->dec_P <= 0)) {
        rc = ensure_host_post(c);                // (lcrc_stage_buffers allocated it: a no-op)
        if (rc) return rc;
        HIP_TRY(c, hipMemcpyAsync(c->h_post, c->d_post, (size_t)n * O * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    }
    HIP_TRY(c, wait_stream(c));
    return LCRC_OK;
}

int lcrc_posteriors_device(lcrc_ctx *c, const float *d_mel, const int *d_off, int n_utts, int n_rows,
                           float *d_post, void *hip_stream)
{
    if (!c) return LCRC_E_ARG;
    if (n_rows < 0 || n_utts < 1 || (n_rows > 0 && (!d_mel || !d_post)) || (n_utts > 1 && !d_off))
        return fail(c, LCRC_E_ARG, "lcrc_posteriors_device: bad argument");
    if (n_rows == 0) return LCRC_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    hipStream_t s = static_cast<hipStream_t>(hip_stream);   // NULL = HIP's default stream
    const int rc = launch(c, d_mel, d_off, n_utts, n_rows, d_post, s, nullptr);
    c->kdone_armed = false;          // asynchronous entry: nobody waits here, lcrc_set_kernel_done_callback does not apply
    return rc;
}

// ---- waveform entry: GPU mel-bank front-end ------------------------------------------------

int lcrc_set_launch_order(lcrc_ctx *c, int ordered)
{
    if (!c) return LCRC_E_ARG;
    gate_membership(c, ordered != 0);
    return LCRC_OK;
}

int lcrc_set_tile_frames(lcrc_ctx *c, int frames)
{
    if (!c) return LCRC_E_ARG;
    if (frames != 0 && frames != 16 && frames != 32) return fail(c, LCRC_E_ARG, "lcrc_set_tile_frames: 0, 16 or 32");
    c->tile_frames = frames;
    return LCRC_OK;
}

int lcrc_output_configure(lcrc_ctx *c, const lcrc_softening *stages, int n_stages, int big_endian)
{
    if (!c) return LCRC_E_ARG;
    if (n_stages < 0 || n_stages > 2 || (n_stages > 0 && !stages)) return fail(c, LCRC_E_ARG, "lcrc_output_configure: 0..2 stages");
    for (int i = 0; i < n_stages; i++)
        if (stages[i].func < LCRC_SOFT_NONE || stages[i].func > LCRC_SOFT_GMM_BYPASS)
            return fail(c, LCRC_E_ARG, "lcrc_output_configure: unknown softening function");
    for (int i = 0; i < 2; i++) c->soft[i] = i < n_stages ? stages[i] : lcrc_softening{0, 0, 0, 0};
    c->out_be = big_endian ? 1 : 0;
    return LCRC_OK;
}

int lcrc_reset(lcrc_ctx *c)
{
    if (!c) return LCRC_E_ARG;
    c->hist_init = false;
    return LCRC_OK;
}

int lcrc_delay(const lcrc_ctx *c) { return c ? c->delay : LCRC_E_ARG; }

// Room for `n` more rows behind the history in the pinned strip, and for n rows of output.
static int ensure_ring(lcrc_ctx *c, size_t n)
{
    const size_t nb = c->nbanks, H = (size_t)c->trap_len - 1, O = c->nets[2].n_out;
    if (c->ring_rows + n > c->ring_cap) {
        if (H + n + n / 2 + 256 > c->ring_cap) {            // grow (keeps the history)
            const size_t cap = std::max<size_t>(4096, 2 * (H + n) + 256);
            float *h = nullptr, *d = nullptr;
            HIP_TRY(c, hipHostMalloc((void **)&h, cap * nb * sizeof(float), kPinnedMapped));
            if (hipHostGetDevicePointer((void **)&d, h, 0) != hipSuccess) { (void)pinned_free(h); return fail(c, LCRC_E_DEVICE, "hipHostGetDevicePointer failed"); }
            if (c->h_ring) {
                const size_t keep = std::min(c->ring_rows, H);
                memcpy(h, c->h_ring + (c->ring_rows - keep) * nb, keep * nb * sizeof(float));
                (void)pinned_free(c->h_ring);
                c->ring_rows = keep;
            }
            c->h_ring = h; c->d_ring = d; c->ring_cap = cap;
        } else {                                             // wrap: the history moves to the front
            // (no kernel reads the strip now: every push that launches also waits for its kernel)
            memmove(c->h_ring, c->h_ring + (c->ring_rows - H) * nb, H * nb * sizeof(float));
            c->ring_rows = H;
        }
    }
    if (n > c->pushout_cap) {
        const size_t cap = n + n / 4 + 64;
        if (c->h_pushout) (void)pinned_free(c->h_pushout);
        c->h_pushout = c->d_pushout = nullptr; c->pushout_cap = 0;
        HIP_TRY(c, hipHostMalloc((void **)&c->h_pushout, cap * O * sizeof(float), kPinnedMapped));
        if (hipHostGetDevicePointer((void **)&c->d_pushout, c->h_pushout, 0) != hipSuccess) return fail(c, LCRC_E_DEVICE, "hipHostGetDevicePointer failed");
        c->pushout_cap = cap;
    }
    return LCRC_OK;
}

// Traps::CalcFeaturesBunched.  The reference slides a 31-slot window one frame at a
// time (traps.cpp:180-219) and evaluates the nets on what the window holds after
// each push.  Here the pushed frames are appended to a strip whose preceding 30 rows are
// the history: row 15+i of the strip [history | pushed] has the window [i, i+30], i.e. exactly
// the ring contents after push i, so ONE launch on the row range [15, 15+n) of that strip
// (30 rows of context, n rows computed) gives the n estimates.  The strip and the output are pinned
// host memory the kernel reads / writes in place: a push is a host memcpy of the frames, the launch
// (split over many workgroups when n is small), one stream wait and a memcpy of the posteriors.
int lcrc_push(lcrc_ctx *c, const float *mel, int n, float *post, int needed)
{
    if (!c) return LCRC_E_ARG;
    if (n < 0 || (n > 0 && !mel) || (n > 0 && needed && !post)) return fail(c, LCRC_E_ARG, "lcrc_push: bad argument");
    if (n == 0) return LCRC_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    const size_t nb = c->nbanks, H = (size_t)c->trap_len - 1, O = c->nets[2].n_out;
    if (!c->hist_init) c->ring_rows = 0;
    // Very large pushes (the offline main call of srec.cpp:1048 hands over a whole utterance) go through the device staging
    // buffers -- [history | frames] copied at PCIe rate, the kernel on HBM -- instead of the kernel reading the frames from
    // and writing the posteriors to mapped host memory row by row; the strip then only keeps the new history.  Up to a few
    // thousand frames the mapped strip wins (512 frames: 70 us against 81 us with the two explicit copies).
    constexpr int kPushStagedMin = 4096;
    if (needed && c->system == SYS_LCRC && n >= kPushStagedMin && c->hist_init) {
        int rc = ensure_staging(c, H + (size_t)n, 1);
        if (rc) return rc;
        memcpy(c->h_mel, c->h_ring + (c->ring_rows - H) * nb, H * nb * sizeof(float));
        memcpy(c->h_mel + H * nb, mel, (size_t)n * nb * sizeof(float));
        HIP_TRY(c, hipMemcpyAsync(c->d_mel, c->h_mel, (H + (size_t)n) * nb * sizeof(float), hipMemcpyHostToDevice, c->stream));
        rc = launch(c, c->d_mel, nullptr, 1, (int)(H + n), c->d_post, c->stream, nullptr, kShift, n, false);
        if (rc) return rc;
        rc = ensure_host_post(c);
        if (rc) return rc;
        HIP_TRY(c, hipMemcpyAsync(c->h_post, c->d_post, (size_t)n * O * sizeof(float), hipMemcpyDeviceToHost, c->stream));
        // the new history while the copies and the kernel run: the last 30 pushed frames
        memcpy(c->h_ring, mel + ((size_t)n - H) * nb, H * nb * sizeof(float));
        c->ring_rows = H;
        HIP_TRY(c, wait_stream(c));
        memcpy(post, c->h_post, (size_t)n * O * sizeof(float));
        c->delay += n;
        if (c->delay > 9999) c->delay = 9999;
        return LCRC_OK;
    }
    int rc = ensure_ring(c, (c->hist_init ? 0 : H) + (size_t)n);
    if (rc) return rc;
    if (!c->hist_init) {                     // first frame floods the history (traps.cpp:184-200)
        for (size_t i = 0; i < H; i++) memcpy(c->h_ring + i * nb, mel, nb * sizeof(float));
        c->ring_rows = H;
    }
    const size_t first = c->ring_rows - H;   // strip = rows [first, first + H + n)
    memcpy(c->h_ring + c->ring_rows * nb, mel, (size_t)n * nb * sizeof(float));
    c->ring_rows += n;
    if (needed) {
        // (the streaming form never decodes: a chunk is not an utterance)
        if (c->system == SYS_LCRC) {
            rc = launch(c, c->d_ring + first * nb, nullptr, 1, (int)(H + n), c->d_pushout, c->stream, nullptr, kShift, n, false);
            if (rc) return rc;
            HIP_TRY(c, wait_stream(c));
            memcpy(post, c->h_pushout, (size_t)n * O * sizeof(float));
        } else {                             // the unfused systems compute the whole strip
            rc = run_host(c, c->h_ring + first * nb, nullptr, 1, (int)(H + n), nullptr, nullptr, false);
            if (rc) return rc;
            // (row of the first pushed frame's window: the output frame's tap, 15 at the usual length)
            const size_t back = (size_t)(c->trap_len - 1 - (c->trap_len - 1) / 2);
            memcpy(post, c->h_post + back * O, (size_t)n * O * sizeof(float));
        }
    }
    if (!c->hist_init) { c->hist_init = true; c->delay = n - 1; }
    else c->delay += n;
    if (c->delay > 9999) c->delay = 9999;
    return LCRC_OK;
}

// Rows [row_first, row_first + row_count) of a strip of n_rows frames (one utterance, or a chunk of one with
// its halos): the other rows are context only.
int lcrc_posteriors_rows(lcrc_ctx *c, const float *mel, int n_rows, int row_first, int row_count, float *post)
{
    if (!c) return LCRC_E_ARG;
    if (n_rows < 0 || row_first < 0 || row_count < 0 || row_first + (long long)row_count > n_rows ||
        (row_count > 0 && (!mel || !post)))
        return fail(c, LCRC_E_ARG, "lcrc_posteriors_rows: bad argument");
    if (row_count == 0) return LCRC_OK;
    if (c->system != SYS_LCRC) return fail(c, LCRC_E_UNSUPPORTED, "row ranges exist for posteriors/system=LCRC only");
    HIP_TRY(c, hipSetDevice(c->device));
    const size_t nb = c->nbanks, O = c->nets[2].n_out;
    int rc = ensure_staging(c, n_rows, 1);
    if (rc) return rc;
    memcpy(c->h_mel, mel, (size_t)n_rows * nb * sizeof(float));
    HIP_TRY(c, hipMemcpyAsync(c->d_mel, c->h_mel, (size_t)n_rows * nb * sizeof(float), hipMemcpyHostToDevice, c->stream));
    rc = launch(c, c->d_mel, nullptr, 1, n_rows, c->d_post, c->stream, nullptr, row_first, row_count);
    if (rc) return rc;
    rc = ensure_host_post(c);
    if (rc) return rc;
    HIP_TRY(c, hipMemcpyAsync(c->h_post, c->d_post, (size_t)row_count * O * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, wait_stream(c));
    memcpy(post, c->h_post, (size_t)row_count * O * sizeof(float));
    return LCRC_OK;
}

int lcrc_debug_fail_alloc(int nth)
{
    const char *e = getenv("LCRC_FAULT_INJECTION");
    if (!e || strcmp(e, "1") != 0) return fail(nullptr, LCRC_E_UNSUPPORTED, "lcrc_debug_fail_alloc: set LCRC_FAULT_INJECTION=1");
    g_fail_alloc = nth;
    return LCRC_OK;
}

int lcrc_debug_fail_launch(int nth)
{
    const char *e = getenv("LCRC_FAULT_INJECTION");
    if (!e || strcmp(e, "1") != 0) return fail(nullptr, LCRC_E_UNSUPPORTED, "lcrc_debug_fail_launch: set LCRC_FAULT_INJECTION=1");
    g_fail_launch = nth;
    return LCRC_OK;
}

int lcrc_set_mean_order(lcrc_ctx *c, int sequential)
{
    if (!c) return LCRC_E_ARG;
    c->mean_sequential = sequential != 0;
    return LCRC_OK;
}

int lcrc_set_hidden_split(lcrc_ctx *c, int workgroups_per_tile)
{
    if (!c) return LCRC_E_ARG;
    if (workgroups_per_tile < 0 || workgroups_per_tile > 64) return fail(c, LCRC_E_ARG, "lcrc_set_hidden_split: 0 (automatic), 1 (never) .. 64");
    c->split_hint = workgroups_per_tile;
    return LCRC_OK;
}

int lcrc_set_arithmetic(lcrc_ctx *c, int arithmetic)
{
    if (!c) return LCRC_E_ARG;
    if (arithmetic != LCRC_ARITH_F32 && arithmetic != LCRC_ARITH_SPLIT_F16)
        return fail(c, LCRC_E_ARG, "lcrc_set_arithmetic: LCRC_ARITH_F32 or LCRC_ARITH_SPLIT_F16");
    if (arithmetic == LCRC_ARITH_SPLIT_F16) {
        if (c->system != SYS_LCRC || !lcrc_has_split_f16(c->nets))
            return fail(c, LCRC_E_UNSUPPORTED, "lcrc_set_arithmetic: split-f16 kernels exist for the shipped LCRC shapes");
        HIP_TRY(c, hipSetDevice(c->device));
        const int rc = ensure_split_f16(c);
        if (rc == LCRC_E_UNSUPPORTED)
            return fail(c, rc, "lcrc_set_arithmetic: the model has no split-f16 form (non-finite weights or biases)");
        if (rc) return rc;
    }
    c->arith = arithmetic;
    return LCRC_OK;
}

#ifdef LCRC_STAMPS
// diagnostic build only (not declared in include/lcrc.h): phase stamps buffer, device pointer
int lcrc_debug_set_stamps(lcrc_ctx *c, void *d_buf)
{
    if (!c) return LCRC_E_ARG;
    c->d_stamps = static_cast<unsigned long long *>(d_buf);
    return LCRC_OK;
}
#endif

int lcrc_set_kernel_done_callback(lcrc_ctx *c, lcrc_kernel_done_fn fn, void *arg)
{
    if (!c) return LCRC_E_ARG;
    c->kdone_fn = fn;
    c->kdone_arg = arg;
    c->kdone_armed = false;
    return LCRC_OK;
}

int lcrc_set_wait_mode(lcrc_ctx *c, int poll_interval_us)
{
    if (!c || poll_interval_us < 0 || poll_interval_us > 100000) return fail(c, LCRC_E_ARG, "lcrc_set_wait_mode: 0 (spin) or a polling interval of 1..100000 us");
    c->poll_wait_us = poll_interval_us;
    return LCRC_OK;
}

int lcrc_set_timing(lcrc_ctx *c, int enabled)
{
    if (!c) return LCRC_E_ARG;
    c->timing = enabled != 0;
    return LCRC_OK;
}

int lcrc_last_kernel_ms(lcrc_ctx *c, float *ms)
{
    if (!c || !ms) return LCRC_E_ARG;
    if (!c->timed) return fail(c, LCRC_E_ARG, "lcrc_last_kernel_ms: no timed launch yet");
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipEventSynchronize(c->ev1));
    HIP_TRY(c, hipEventElapsedTime(ms, c->ev0, c->ev1));
    return LCRC_OK;
}

}  // extern "C"
