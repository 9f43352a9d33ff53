// Host Viterbi (phnrec_amd/csrc/host/phndec.cpp) reading DRAM-cold posteriors WHILE other threads refill the same
// buffers at a set rate: a rehearsal, on the host alone, of what N GPUs' posterior stores do to the decoder threads
// (`-F` / `-E` without `-D`: every GPU stores 744 B per frame -- 22.8 GB/s at 30.6 M frames/s -- into pinned host memory
// and the pool's Viterbi threads read it back out of DRAM).  The writers are CPU threads issuing non-temporal 64-byte
// stores (write-combining, past the caches, as a device's PCIe writes arrive on this platform); they stand in for the
// GPUs -- no GPU is touched.  What the record shows: Viterbi CPU nanoseconds per frame (CLOCK_THREAD_CPUTIME_ID, as the
// CLI's PHNREC_STATS counts them) without writers and under each write rate the box's cores can produce.
//   host_mem_load [viterbi threads = 8] [writer threads = 8] [seconds per point = 2] [rates in GB/s ... ; -1 = unthrottled]
// g++ -O2 -std=c++17 -mavx512f (see Makefile)
#include <immintrin.h>
#include <sys/mman.h>

#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <random>
#include <thread>
#include <vector>

#include "phndec.h"

namespace {

constexpr int kP = 61, kCols = 186, kT = 900;       // HU: 61 phonemes x 3 states + 3 = 186 outputs; configs[3]'s mean file length
constexpr int kRowsPerBuf = 32768;                  // one launch's posteriors (-b default)
constexpr int kBufs = 24;                           // 8 GPUs x 3 contexts

long long ThreadCpuNs()
{
    timespec ts;
    clock_gettime(CLOCK_THREAD_CPUTIME_ID, &ts);
    return (long long)ts.tv_sec * 1000000000LL + ts.tv_nsec;
}

double Now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// non-temporal copy of `bytes` (multiple of 64) from a cache-resident template
void StreamCopy(float *dst, const float *src, size_t bytes)
{
#ifdef __AVX512F__
    for (size_t i = 0; i < bytes / 64; i++) _mm512_stream_ps(dst + i * 16, _mm512_load_ps(src + i * 16));
#else
    for (size_t i = 0; i < bytes / 32; i++) _mm256_stream_ps(dst + i * 8, _mm256_load_ps(src + i * 8));
#endif
    _mm_sfence();
}

}  // namespace

int main(int argc, char **argv)
{
    const int nv = argc > 1 ? atoi(argv[1]) : 8, nw = argc > 2 ? atoi(argv[2]) : 8;
    const double secs = argc > 3 ? atof(argv[3]) : 2.0;
    std::vector<double> rates;
    for (int i = 4; i < argc; i++) rates.push_back(atof(argv[i]));
    if (rates.empty()) rates = {0, 22.8, 45.6, 91.2, 182.4, -1};

    // one utterance of valid log-posteriors (peaky, slowly moving), 64-byte aligned: the template every buffer is filled from
    const size_t utt_floats = (size_t)kT * kCols, utt_bytes = utt_floats * 4 / 64 * 64;
    float *tmpl = static_cast<float *>(aligned_alloc(64, utt_bytes + 64));
    {
        std::mt19937 g(5);
        int cur = 0;
        for (int t = 0; t < kT; t++) {
            if (g() % 7 == 0) cur = g() % kP;
            float sum = 0, row[kCols];
            for (int c = 0; c < kCols; c++) {
                row[c] = expf((c / 3 == cur ? 8.0f : 0.0f) + std::uniform_real_distribution<float>(0, 2)(g));
                sum += row[c];
            }
            for (int c = 0; c < kCols; c++) tmpl[(size_t)t * kCols + c] = logf(row[c] / sum);
        }
    }
    const int utts_per_buf = kRowsPerBuf / kT;
    const size_t buf_bytes = (size_t)utts_per_buf * utt_bytes;
    std::vector<float *> bufs;
    for (int b = 0; b < kBufs; b++) {
        void *p = mmap(nullptr, buf_bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        if (p == MAP_FAILED) { perror("mmap"); return 1; }
        (void)mlock(p, buf_bytes);                                  // pinned, like the contexts' buffers (best effort)
        bufs.push_back(static_cast<float *>(p));
        for (int u = 0; u < utts_per_buf; u++) memcpy(bufs.back() + (size_t)u * (utt_bytes / 4), tmpl, utt_bytes);
    }
    std::vector<std::string> names;
    for (int i = 0; i < kP; i++) names.push_back("p" + std::to_string(i));
    printf("host_mem_load: %d Viterbi threads (P=%d, %d columns, utterances of %d frames, prefetch 8 frames ahead as the CLI's "
           "Stage3), %d writer threads (non-temporal stores), %d buffers of %.1f MB, %.1f s per point\n",
           nv, kP, kCols, kT, nw, kBufs, buf_bytes / 1e6, secs);
    printf("%12s %14s %16s %16s %12s\n", "asked GB/s", "written GB/s", "Viterbi ns/frame", "M frames/s wall", "labels/utt");

    bool warm = true;                               // one short unreported point first (thread arenas, page tables, clocks)
    rates.insert(rates.begin(), 0.0);
    for (double rate : rates) {
        std::atomic<bool> stop(false);
        std::atomic<long long> written(0), frames(0), cpu_ns(0), labels(0), utts(0);
        std::vector<std::thread> th;
        const int writers = rate == 0 ? 0 : nw;
        const double t_start = Now();
        for (int w = 0; w < writers; w++)
            th.emplace_back([&, w] {
                const double per_thread = rate > 0 ? rate * 1e9 / writers : 0;       // bytes per second of this writer
                long long mine = 0;
                size_t k = (size_t)w * 7919;
                while (!stop.load(std::memory_order_relaxed)) {
                    float *dst = bufs[k % kBufs] + (size_t)((k / kBufs) % utts_per_buf) * (utt_bytes / 4);
                    StreamCopy(dst, tmpl, utt_bytes);
                    k += (size_t)writers;
                    mine += (long long)utt_bytes;
                    if (per_thread > 0)
                        while (!stop.load(std::memory_order_relaxed) && mine > (Now() - t_start) * per_thread) _mm_pause();
                }
                written += mine;
            });
        for (int v = 0; v < nv; v++)
            th.emplace_back([&, v] {
                size_t k = (size_t)v * 104729 + 13;
                long long fr = 0, ns = 0, nl = 0, nu = 0;
                phnrec::PhnDec d;
                d.SetPhonemes(names);
                d.SetStatesPerPhn(3);
                d.SetTimePruning(40);
                d.SetWPenalty(-2.8125f);
                while (!stop.load(std::memory_order_relaxed)) {
                    // an utterance some writer passed a while ago: a different buffer each time, far from this core's caches
                    const float *post = bufs[k % kBufs] + (size_t)((k / kBufs) % utts_per_buf) * (utt_bytes / 4);
                    k += (size_t)nv * 31;
                    d.Init();
                    const long long a = ThreadCpuNs();
                    const size_t row_bytes = (size_t)kCols * 4;
                    const char *base = reinterpret_cast<const char *>(post), *end = base + (size_t)kT * row_bytes;
                    for (int r = 0; r < kT; r++) {
                        const char *q = base + (size_t)(r + 8) * row_bytes;
                        for (const char *e = q + row_bytes; q < e && q < end; q += 64) __builtin_prefetch(q, 0, 3);
                        d.ProcessFrame(post + (size_t)r * kCols);
                    }
                    d.Done();
                    ns += ThreadCpuNs() - a;
                    fr += kT;
                    nl += (long long)d.Labels().size();
                    nu++;
                }
                frames += fr; cpu_ns += ns; labels += nl; utts += nu;
            });
        std::this_thread::sleep_for(std::chrono::duration<double>(warm ? 0.3 : secs));
        stop = true;
        for (auto &t : th) t.join();
        const double dt = Now() - t_start;
        if (warm) { warm = false; continue; }
        printf("%12s %14.1f %16.1f %16.2f %12.1f\n", rate < 0 ? "unthrottled" : std::to_string(rate).substr(0, 6).c_str(),
               written.load() / dt / 1e9, frames.load() ? (double)cpu_ns.load() / frames.load() : 0.0, frames.load() / dt / 1e6,
               utts.load() ? (double)labels.load() / utts.load() : 0.0);
        fflush(stdout);
    }
    return 0;
}
