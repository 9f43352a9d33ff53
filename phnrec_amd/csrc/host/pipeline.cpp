// pipeline.cpp -- SpeechRec::RunPipeline: a file list (or one file) from its first line to its last as one pipeline of
// feeder, host pool, per-context GPU workers and in-order writer; ProcessFileList / ProcessFileListLine on top of it.
// (The per-utterance stages it calls -- Stage1, Stage3, EmitLabels, ParseLine -- and the set-up of contexts: srec.cpp.)
#include "srec.h"
#include "veclog.h"

#include <fcntl.h>
#include <sched.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <cerrno>
#include <chrono>
#include <cmath>
#include <cctype>
#include <cstring>
#include <ctime>
#include <functional>
#include <map>
#include <memory>
#include <set>
#include <thread>

#include "htk.h"

namespace phnrec {

// posteriors/softening_func and decoder/softening_func as the device evaluates them (lcrc_output_configure)
static lcrc_softening DeviceSoftening(const std::string &f, const float *a)
{
    lcrc_softening s = {LCRC_SOFT_NONE, a[0], a[1], a[2]};
    if (f == "log") s.func = LCRC_SOFT_LOG;
    else if (f == "igor") s.func = LCRC_SOFT_IGOR;
    else if (f == "gmm_bypass") s.func = LCRC_SOFT_GMM_BYPASS;
    return s;
}

// A file list (or the one file of -i) runs as ONE pipeline from its first line to its last:
//   feeder (the calling thread)  parses lines and queues stage 1 of each job on the pool: file read [+ host
//                                front-end + sentence norm]; with -F only a stat() -- the files are then read
//                                straight into the launching context's pinned byte buffer
//   GPU workers (one thread per context, three contexts per GPU) each take the next launch -- the longest run of
//                                consecutive staged jobs within batch_frames_ --, gather, run the kernel(s), and
//                                decode / dump the launch's utterances on the pool (chunks of a blocked worker run
//                                ahead of queued stage-1 tasks)
//   writer (whoever finishes a job) MLF entries leave in list order.
// Nothing joins the contexts before the end of the list; the window of jobs in flight is bounded by frames
// (a few launches per context) and bytes, not by a file count.  Launch boundaries depend on timing only at the
// very end of a list; results do not depend on them (the CLI pins the fused kernel: batch-invariant bits).
// Errors keep the reference's sequential meaning (srec.cpp:1246-1290): a file that cannot be read or a bad list
// line stops the run THERE -- everything before it is computed and written, then the error is reported.
namespace {

// A whole file into `dst` (exactly `bytes` of it): one open, reads until done, close -- no stdio buffer in between
// (a FILE's 4 KiB buffer would copy every byte twice; the files go straight into pinned memory)
bool ReadWholeFile(const char *path, unsigned char *dst, long long bytes)
{
    const int fd = open(path, O_RDONLY | O_CLOEXEC);
    if (fd < 0) return false;
    long long got = 0;
    while (got < bytes) {
        const ssize_t r = read(fd, dst + got, (size_t)(bytes - got));
        if (r < 0 && errno == EINTR) continue;
        if (r <= 0) break;
        got += r;
    }
    close(fd);
    return got == bytes;
}

// Launch slots of one physical GPU.  Contexts that share a GPU run their launches under processor sharing: three
// equal launches submitted together also END together, their contexts then read / decode together while the GPU has
// nothing to do, and the convoy repeats (profiles/r03_cli_timeline.txt: device busy 69-80 % of a list run).  With at
// most `slots` launches admitted at a time -- first come, first served -- the admitted ones finish one after the other
// and a context that is ready takes the slot the moment one leaves: the contexts fall out of step and stay so.
class DeviceSlots {
public:
    explicit DeviceSlots(int n) : free_(n) {}
    void Acquire()
    {
        std::unique_lock<std::mutex> l(mu_);
        const long long my = next_++;
        cv_.wait(l, [&] { return free_ > 0 && serving_ == my; });
        free_--;
        serving_++;
        cv_.notify_all();
    }
    void Release()
    {
        std::lock_guard<std::mutex> l(mu_);
        free_++;
        cv_.notify_all();
    }

private:
    std::mutex mu_;
    std::condition_variable cv_;
    int free_;
    long long next_ = 0, serving_ = 0;
};

// CPU seconds of the calling thread (not wall clock: a thread that waits for its time slice under a cgroup quota,
// beside spinning waiters, consumes nothing -- the host ceiling wants what the cores must DELIVER)
long long ThreadCpuNs()
{
    timespec ts;
    clock_gettime(CLOCK_THREAD_CPUTIME_ID, &ts);
    return (long long)ts.tv_sec * 1000000000LL + ts.tv_nsec;
}

// CPUs of a GPU's NUMA node -- the node its PCIe root hangs on -- that the process may use.  The thread that feeds and
// waits for GPU g keeps to them (it touches g's pinned staging buffers, allocated next to the GPU, and its doorbells),
// and the pool keeps to the nodes of the GPUs in use.  Only where the node is known and has CPUs inside the process's
// affinity mask; silently nothing otherwise.
bool GpuNodeCpus(int device, cpu_set_t *want)
{
    CPU_ZERO(want);
    char bus[64] = {0};
    if (lcrc_device_pci_bus_id(device, bus, sizeof bus) != 0 || !bus[0]) return false;
    for (char *q = bus; *q; q++) *q = (char)tolower((unsigned char)*q);
    char path[256];
    snprintf(path, sizeof path, "/sys/bus/pci/devices/%s/numa_node", bus);
    int node = -1;
    if (FILE *f = fopen(path, "r")) {
        if (fscanf(f, "%d", &node) != 1) node = -1;
        fclose(f);
    }
    if (node < 0) return false;
    snprintf(path, sizeof path, "/sys/devices/system/node/node%d/cpulist", node);
    FILE *f = fopen(path, "r");
    if (!f) return false;
    char list[4096] = {0};
    const bool got = fgets(list, sizeof list, f) != nullptr;
    fclose(f);
    if (!got) return false;
    cpu_set_t allowed;
    if (sched_getaffinity(0, sizeof allowed, &allowed) != 0) return false;
    for (char *q = list; *q;) {                    // "0-63,128-191"
        char *end = nullptr;
        const long a = strtol(q, &end, 10);
        if (end == q) break;
        long b = a;
        if (*end == '-') { q = end + 1; b = strtol(q, &end, 10); if (end == q) break; }
        for (long c = a; c <= b && c < CPU_SETSIZE; c++)
            if (CPU_ISSET((int)c, &allowed)) CPU_SET((int)c, want);
        if (*end != ',') break;
        q = end + 1;
    }
    return CPU_COUNT(want) > 0;
}

void PinToGpuNode(int device)
{
    cpu_set_t want;
    if (GpuNodeCpus(device, &want)) (void)sched_setaffinity(0, sizeof want, &want);
}

struct Slot {
    int state = 0;          // 0: stage 1 pending, 1: staged (waits for a launch), 2: in a launch, 3: done
};

}  // namespace

bool SpeechRec::RunPipeline(DataFormat in, DataFormat out, const std::function<int(Job &)> &next, FILE *mlf, bool single_file)
{
    using clock = std::chrono::steady_clock;
    const auto t0 = clock::now();
    if (!pool_) {
        const int threads = host_threads_ > 0 ? host_threads_ : UsableCpus();
        pool_.reset(new ThreadPool(threads > 1 ? threads : 0));
    }
    const bool need_gpu = (in == dfWaveform || in == dfParams) && (out == dfPosteriors || out == dfStrings);
    // A list called as the reference is called (no -F, no -E) takes the GPU front-end by itself:
    //   -F (the whole front-end) where its features are the host front-end's bit for bit -- ln() as this host's libm takes it
    //      (LibmLogfForm: glibc) and a configuration that asks for nothing only the host / -E road does (framenorm/*,
    //      sent_max_norm, sent_chmax_norm) -- from ONE GPU on, for lists of at least ~100 files (long_list_): the same bytes
    //      out (tests compare dumps and MLFs byte for byte), the same or a higher rate on one GPU (configs[3] list x 8: 32.3
    //      against 30.9 M frames/s) at an eighth of the host's CPU time (3.6 against 30.6 CPU-s), and the only road on which
    //      the host side feeds more than one GPU (the host front-end's 0.36 us per frame and core feeds 1.2 on 16 cores);
    //   -E (the energies; ln() and normalisations on the host: the same bits on any libm) from TWO GPUs on otherwise;
    // where the GPU front-end takes the configuration at all (GpuFrontendTakesConfig: the limits lcrc_frontend_configure
    // enforces; anything else keeps the host front-end).  One file, and short lists on one GPU, keep the host front-end.
    // PHNREC_NO_AUTO_E=1 keeps it whatever the list and -g say.  The choice lives in auto_frontend_ / auto_energies_ /
    // auto_decoder_: what the caller set (SetGpuFrontend, SetGpuEnergies, SetGpuDecoder) is never overwritten.
    // THE FIRST CALL DECIDES: contexts are configured for a front-end when they are created and live as long as this
    // SpeechRec, so the choice is made once, while no context has been planned, and holds for every later call on the object
    // (a single file first, then a long list: the list keeps the host front-end; the CLI makes one call per process).
    // ModeString() and the default frames per launch report that first choice.
    if (gpus_.empty()) {
        const bool eligible = need_gpu && !single_file && in == dfWaveform && !gpu_frontend_ && !gpu_energies_ &&
                              wave_.noise_level == 0.0f && GpuFrontendTakesConfig() && !getenv("PHNREC_NO_AUTO_E");
        const bool same_bits = eligible && LibmLogfForm() != 0 && C.GetFloat("framenorm", "shift") == 0.0f &&
                               C.GetFloat("framenorm", "min_floor") == -9999.9f && !sent_max_norm_ && !sent_chmax_norm_;
        auto_frontend_ = same_bits && (n_gpus_ >= 2 || long_list_);
        auto_energies_ = eligible && !same_bits && n_gpus_ >= 2;
        // A list over four or more GPUs that ends in labels: the decoder runs on the GPUs too (-D) by itself.  Its labels are
        // the host decoder's bit for bit (tested), it costs a GPU 1-3 % of its rate and takes the Viterbi -- half of what is
        // left of the host's work with -F, a third with -E -- off the cores that eight GPUs' lists otherwise bring to their
        // limit.  PHNREC_NO_AUTO_D=1 keeps the host decoder.
        auto_decoder_ = need_gpu && !single_file && out == dfStrings && !gpu_decoder_ && n_gpus_ >= 4 && !phn_names_.empty() &&
                        phn_names_.size() <= 64 && states_per_phn_ >= 1 && states_per_phn_ <= 4 && time_pruning_ >= 1 &&
                        time_pruning_ <= 63 && !getenv("PHNREC_NO_AUTO_D");
    }
    // frames per launch: 32 768; with the decoder on the GPU 65 536 (one decoder wave per utterance and a launch as long as
    // its longest utterance: twice the utterances per launch, half the launches); -b overrides
    if (!batch_given_) batch_frames_ = need_gpu && DecoderOn() && out == dfStrings ? 65536 : 32768;
    if (need_gpu) {
        if (!traps_enabled_) return Fail("The 'traps' module have to be enabled for generating posteriors\n");
        // a list: three contexts per GPU -- while one's kernel runs, another reads its files and runs its front-end and
        // the third's utterances are decoded (on the host, or by the decoder kernel behind its posterior kernel: that
        // kernel's 2 ms are part of its context's chain).  Measured on the configs[3] list x 4, one GPU (profiles/
        // r05_ab_runs.txt 3): 2 / 3 / 4 contexts -F 30.3 / 32.2 / 32.1, -F -D 24.9 / 31.4 / 29.3, -E -D 22.7 / 26.3 / 26.9 M
        // frames/s -- with two contexts the device has no posterior kernel to run for 16 % of a -F -D list, with three for
        // 0.9 %.  Every context costs ~8 ms of start-up that the HIP runtime serialises (stream, 30 MB of pinned staging):
        // from four GPUs on a list gets three per GPU only if it is long enough to pay for them -- 128 KB of list file per
        // GPU, some 4000 files, a tenth of a second of work per GPU --, otherwise two.
        const bool worth_three = std::max(1, n_gpus_) < 4 || list_bytes_ / std::max(1, n_gpus_) >= (128 << 10);
        const int per_gpu = single_file ? 1 : worth_three ? 3 : 2;
        // (only the places: each context is built by its own worker below, beside the running list)
        if (!PlanGpus(per_gpu)) return false;
        // (which of glibc's two logf sequences this host's libm runs -- 300 000 probes, ~12 ms -- is found out HERE, on this
        //  thread, while the HIP runtime starts up on others; the first context's set-up asked in front of its first launch)
        if (FrontendOn()) (void)LibmLogfForm();
        fe_vector_size_ = C.GetInt("melbanks", "vector_size");
        fe_vector_step_ = std::max(1, C.GetInt("melbanks", "vector_step"));
    }
    // (a model the device decoder does not take -- more states than posterior outputs, say -- keeps the host decoder
    //  when -D was this function's own idea)
    if (auto_decoder_ && (int)phn_names_.size() * states_per_phn_ > n_out_) {
        auto_decoder_ = false;
        if (!batch_given_ && !gpu_decoder_) batch_frames_ = 32768;
    }
    const bool dev_dec = need_gpu && DecoderOn() && out == dfStrings;
    // With the decoder on the device a GPU's contexts run their posterior kernels one after the other, in the order their
    // launches were queued (lcrc_set_launch_order), instead of sharing the device: two that share it end together -- and then
    // decode together while the device has no posterior kernel to run.  Measured on the configs[3] list x 4, one GPU, three
    // contexts (profiles/r05_ab_runs.txt 3): -F -D 28.5 -> 30.9, -E -D 23.3 -> 27.7 M frames/s.  Without the decoder the
    // launches are half as long and their kernels store 744 B per frame into pinned HOST memory while they run: there two
    // kernels sharing the device came out ahead (-F 31.3 against 29.4 in order) and those modes keep the shared device.
    // PHNREC_LAUNCH_ORDER=0/1 overrides (experiments).
    bool launch_order = dev_dec && !single_file;
    if (const char *e = getenv("PHNREC_LAUNCH_ORDER")) launch_order = need_gpu && !single_file && atoi(e) != 0;
    // Two contexts per GPU (a short list over four or more GPUs): the decoder kernel of a launch (2 ms, as long as its longest
    // utterance) runs beside the NEXT launch's kernels of the same context instead of in front of them (lcrc_set_decoder_overlap:
    // two sets of posterior / label buffers per context alternate) -- with two contexts the decoder's latency is otherwise
    // what the device waits for (-F -D 25.6 -> 27.8, -E -D 22.7 -> 23.7); three contexts hide it by themselves and do better
    // without (31.0 against 27.3, 27.7 against 25.6).
    // PHNREC_DECODER_OVERLAP=0/1 overrides (experiments).
    bool dec_overlap = dev_dec && !single_file && (int)gpus_.size() <= 2 * std::max(1, n_gpus_);
    if (const char *e = getenv("PHNREC_DECODER_OVERLAP")) dec_overlap = dev_dec && !single_file && atoi(e) != 0;
    // -v: one line that names the road this run takes through the device -- the same command line takes different ones by
    // list length, -g, the host's libm and the configuration, all with the same output bytes
    if (verbose_ && need_gpu) {
        const int n = std::max(1, n_gpus_);
        std::string fe = FrontendOn() ? (auto_frontend_ ? "GPU (-F, chosen by itself: a list of ~100 files or more / several GPUs, this libm's logf() "
                                                          "reproduced on the device)" : "GPU (-F)")
                         : EnergiesOn() ? (auto_energies_ ? "GPU up to the mel-bank energies, ln() and normalisations on the host (-E, chosen by itself: "
                                                            "several GPUs; -F not bit-exact for this libm / configuration)" : "GPU up to the mel-bank energies (-E)")
                         : in == dfWaveform ? (single_file ? "host (one file)" : long_list_ || n >= 2 ? "host (PHNREC_NO_AUTO_E, source/noise_level or a configuration the GPU front-end does not take)"
                                                                                                      : "host (a short list on one GPU)")
                                            : "none (parameter files in)";
        std::string dec = out != dfStrings ? "none (posterior dump)"
                          : dev_dec ? (auto_decoder_ ? "GPU (-D, chosen by itself: four or more GPUs)" : "GPU (-D)") : "host";
        char line[900];
        snprintf(line, sizeof line,
                 "Device path: front-end %s; decoder %s; %d GPU(s) x %d context(s) planned%s; posterior kernels of a GPU's contexts %s; "
                 "%d frames per launch%s\n",
                 fe.c_str(), dec.c_str(), n, (int)gpus_.size() / n,
                 single_file ? "" : " (contexts that share a model come up beside the running list, only while it is long enough)",
                 launch_order ? "one after the other in queueing order" : "share the device",
                 batch_frames_, dev_dec && dec_overlap ? "; a launch's decoder runs beside the context's next launch" : "");
        Log(line);
    }
    std::vector<std::string> phn_names;
    ContextPlan plan;
    if (need_gpu) {
        // posterior writer path: both softening functions and the dump's byte order run in the posterior
        // kernel's epilogue; the host only decodes or writes
        plan.soft[0] = DeviceSoftening(post_soft_, post_soft_arg_);
        plan.soft[1] = DeviceSoftening(dec_soft_, dec_soft_arg_);
        plan.n_soft = out == dfStrings ? 2 : 1;
        plan.big_endian = out == dfPosteriors;
        // -D: the decoder runs behind the posterior kernel and only labels cross PCIe
        if (dev_dec) phn_names = phn_names_;
        plan.device_decoder = dev_dec;
        plan.decoder_overlap = dec_overlap;
        plan.launch_order = launch_order;
        // Every context has a thread waiting for it.  Spinning (the default) is the fastest way to notice a finished
        // launch and costs a core each: fine for one GPU's three on 16 cores (sleeping waits lose 15 % with -F there,
        // profiles/r03_ab_runs.txt 14), not when the waiting threads of many GPUs would take more than half of the cores
        // the front-end and the decoder of the same run need (16 contexts on 16 cores: +10-20 % with sleeping waits,
        // item 18).  PHNREC_WAIT_POLL_US overrides (0 = spin).
        plan.poll_us = (int)gpus_.size() * 2 > UsableCpus() ? 50 : 0;
        if (const char *e = getenv("PHNREC_WAIT_POLL_US")) plan.poll_us = std::max(0, atoi(e));
        // A list: every context's buffers for launches of batch_frames_, allocated by the context's worker before its first
        // launch instead of inside it (pinning 30 MB of posterior buffer is 5-10 ms; the other contexts run meanwhile).
        // (lists of at least ~100 entries, by the size of the list file: a short list's launches never fill such buffers and
        //  its few files are done sooner than 3 x 40 MB are pinned)
        if (!single_file && long_list_) {
            // (a -b beyond 131 072 frames is reserved up to that: a short list would never fill the rest, the buffers grow
            //  on demand as before)
            plan.reserve_rows = std::min(batch_frames_, 131072);
            if ((FrontendOn() || EnergiesOn()) && in == dfWaveform)
                plan.reserve_wave_bytes = ((long long)plan.reserve_rows * C.GetInt("melbanks", "vector_step") + C.GetInt("melbanks", "vector_size")) *
                                          (wave_.format == WF_LIN16 ? 2 : 1) + 4096;
        }
    }
    // (what is left in front of the list: the pool and the plan -- the contexts come up beside it, inside `seconds`)
    const auto t1 = clock::now();
    stats_.init_seconds += std::chrono::duration<double>(t1 - t0).count();

    struct Item { Job job; Slot slot; long long seq = 0; };
    std::mutex mu;
    std::condition_variable cv_feed, cv_work, cv_idle;
    std::deque<std::unique_ptr<Item>> win;     // jobs in flight, list order; win.front() has sequence number `base`
    long long base = 0, next_launch = 0;       // next_launch: first job no launch has taken yet
    int pending1 = 0;                          // stage-1 tasks queued or running
    long long staged_frames = 0, staged_bytes = 0;   // staged, not yet taken by a launch
    bool eof = false, feeder_blocked = false;
    long long stop_seq = -1;                   // first job whose stage 1 failed: nothing at or behind it is launched
    std::string fatal;                         // GPU / output error: the run ends
    bool fatal_pending_clone = false;          // a clone gave up because its GPU's first context failed (whose message is `fatal`)
    long long files_done = 0, frames_done = 0;
    // CPU time the host stages take, summed over the threads that run them (what the cores must deliver however
    // fast the GPUs are: PHNREC_STATS prints it, bench.py derives the host ceiling of a list from it)
    std::atomic<long long> stage1_us(0), read_us(0), gather_us(0), stage3_us(0), stage1_jobs(0);
    struct CpuTimer {
        std::atomic<long long> &acc;
        long long t0;
        explicit CpuTimer(std::atomic<long long> &a) : acc(a), t0(ThreadCpuNs()) {}
        ~CpuTimer() { acc += ThreadCpuNs() - t0; }
    };
    std::atomic<bool> first_launch(true);
    // PHNREC_TRACE_PIPELINE=1: a time line of the workers' steps on stderr at the end of the run (diagnostic)
    const bool trace_on = getenv("PHNREC_TRACE_PIPELINE") != nullptr;
    std::mutex trace_mu;
    std::vector<std::string> trace_rows;
    auto trace = [&](int ctx, const char *what, long long a = 0, long long b = 0) {
        if (!trace_on) return;
        char line[240];
        long rss_pages = 0;                              // the process's resident set as the step ends (/proc/self/statm)
        if (strncmp(what, "ctx:", 4) == 0 || strncmp(what, "worker", 6) == 0)
            if (FILE *f = fopen("/proc/self/statm", "r")) { long sz = 0; if (fscanf(f, "%ld %ld", &sz, &rss_pages) != 2) rss_pages = 0; fclose(f); }
        snprintf(line, sizeof line, "%9.3f ms  (%lld us, thread %ld)  ctx %d  %-18s %lld %lld%s\n",
                 std::chrono::duration<double, std::milli>(clock::now() - t1).count(),
                 (long long)std::chrono::duration<double, std::micro>(clock::now().time_since_epoch()).count() % 100000000LL,
                 (long)(size_t)pthread_self() % 1000, ctx, what, a, b,
                 rss_pages ? ("  rss " + std::to_string(rss_pages * 4 / 1024) + " MB").c_str() : "");
        std::lock_guard<std::mutex> l(trace_mu);
        trace_rows.emplace_back(line);
    };
    const int n_ctx = need_gpu ? (int)gpus_.size() : 0;
    const int max_pending = std::max(8, 4 * std::max(1, pool_->Size()));
    const long long max_staged_frames = (long long)batch_frames_ * (n_ctx + 2);
    const long long max_staged_bytes = 2LL << 30;
    const long long start_up_bytes = 512LL << 20;
    const size_t max_window = 1 << 16;

    // in-order output; call with `mu` held
    auto drain = [&]() {
        while (!win.empty() && win.front()->slot.state == 3 && fatal.empty()) {
            Job &j = win.front()->job;
            if (!j.ok) { fatal = j.err; break; }
            if (mlf) fputs(j.labels.c_str(), mlf);
            files_done++;
            frames_done += j.frames;
            win.pop_front();
            base++;
        }
        cv_feed.notify_all();
        cv_idle.notify_all();
    };

    auto job_bytes = [](const Job &j) { return (long long)((j.mel.size() + j.post.size()) * sizeof(float)); };
    // stage 1 of a chunk of consecutive jobs, then ONE trip through the pipeline's lock for all of them (with -F a job's
    // stage 1 is a stat(): one task, one wake-up of the waiting workers and one rescan of the window PER FILE capped the
    // pipeline at 80 k files a second whatever the files' length -- 2.5 GPUs' worth of configs[3]'s 894-frame files)
    auto stage1_task = [&](const std::vector<Item *> &its) {
        {
            CpuTimer tm(stage1_us);
            for (Item *it : its) Stage1(in, out, it->job);
        }
        stage1_jobs += (long long)its.size();
        std::lock_guard<std::mutex> l(mu);
        for (Item *it : its) {
            pending1--;
            it->slot.state = 1;
            staged_frames += it->job.frames;
            staged_bytes += job_bytes(it->job);
        }
        cv_work.notify_all();
        cv_feed.notify_all();
        cv_idle.notify_all();
    };

    // The next launch: the longest run of consecutive staged jobs from next_launch within batch_frames_ (at least
    // one job).  Blocks until that run is closed -- full, at the end of the list, or in front of a failed job.
    // false: there will be no more launches.
    auto take_launch = [&](std::vector<Item *> &items) -> bool {
        std::unique_lock<std::mutex> l(mu);
        for (;;) {
            if (!fatal.empty()) return false;
            items.clear();
            long long frames = 0, bytes = 0;
            bool closed = false, final = false;     // final: no job will ever follow this run
            for (long long q = next_launch;; q++) {
                if (stop_seq >= 0 && q >= stop_seq) { closed = final = true; break; }
                // (a feeder that waits for room while nothing is being staged will not extend this run)
                if (q - base >= (long long)win.size()) { final = eof; closed = eof || feeder_blocked; break; }
                Item *it = win[(size_t)(q - base)].get();
                if (it->slot.state == 0) break;
                if (!it->job.ok) { stop_seq = q; closed = final = true; cv_feed.notify_all(); break; }
                if (!items.empty() && frames + it->job.frames > batch_frames_) { closed = true; break; }
                items.push_back(it);
                frames += it->job.frames;
                bytes += job_bytes(it->job);
                if (frames >= batch_frames_ || bytes >= max_staged_bytes) { closed = true; break; }
            }
            if (items.empty() && final) return false;
            if (closed && !items.empty()) {
                for (Item *it : items) {
                    it->slot.state = 2;
                    staged_frames -= it->job.frames;
                    staged_bytes -= job_bytes(it->job);
                }
                next_launch += (long long)items.size();
                cv_feed.notify_all();
                return true;
            }
            cv_work.wait(l);
        }
    };

    // two launch slots per PHYSICAL device (logical GPUs mapped onto one device share its slots)
    const int slots_per_gpu = 2;
    std::map<int, std::unique_ptr<DeviceSlots>> dev_slots;
    if (need_gpu && slots_per_gpu > 0 && !single_file)
        for (int d : gpu_devices_)
            if (!dev_slots.count(d)) dev_slots[d].reset(new DeviceSlots(slots_per_gpu));
    const int n_log = std::max(1, (int)gpu_devices_.size());
    // items per ParallelFor chunk so that a chunk is worth waking a thread for: ~8000 frames of decoding / formatting
    // (a quarter of a millisecond), ~1 MB of file reads
    auto frame_grain = [](int cnt, long long frames) {
        return (int)std::max<long long>(1, 8000LL * cnt / std::max<long long>(1, frames));
    };
    std::vector<double> kms((size_t)n_ctx, 0.0);
    // conversions that do not touch the GPU (-t par, -s post) take the same road -- runs of consecutive staged jobs --
    // so that nothing BEHIND a file that cannot be read is ever written
    auto host_worker = [&]() {
        std::vector<Item *> items;
        while (take_launch(items)) {
            long long fr = 0;
            for (Item *it : items) fr += it->job.frames;
            pool_->ParallelFor((int)items.size(), [&](int k) {
                CpuTimer tm(stage3_us);
                Job &j = items[k]->job;
                Stage3(out, j, mlf != nullptr, out == dfParams ? nullptr : j.post.data(), j.cols);
                std::vector<float>().swap(j.mel);
                std::vector<float>().swap(j.post);
            }, frame_grain((int)items.size(), fr));
            std::lock_guard<std::mutex> l(mu);
            for (Item *it : items) it->slot.state = 3;
            drain();
        }
    };
    std::atomic<int> contexts_up(0), contexts_committed(0);
    std::atomic<bool> pool_placed(false);
    // A context that shares another's weights comes up 20-50 ms after that one (stream, tables, 30-40 MB of pinned staging --
    // steps the runtime serialises against everything else the process does on the GPUs, the running contexts' launches
    // included).  It is worth that only if the list is still long enough by then: more than ~1.5 M frames -- 50 ms of a
    // GPU's work -- left to launch per context already up or on its way.  While the list is still being read its length is
    // unknown and the answer is yes.  (Eight GPUs over configs[3]'s 8.9 M frames: the eight first contexts finish the list
    // in the time sixteen clones would take to come up.)  PHNREC_ALL_CONTEXTS=1: every planned context, always.
    const bool all_contexts = getenv("PHNREC_ALL_CONTEXTS") != nullptr;
    // (the first context of every physical device is always created)
    const int n_first_contexts = (int)std::set<int>(gpu_devices_.begin(), gpu_devices_.end()).size();
    // ... and a PHYSICAL device gets three at most, however many logical GPUs are mapped onto it (PHNREC_DEVICE_MAP=0,0,...):
    // a fourth context on a device adds nothing (profiles/r05_ab_runs.txt 3: 2 / 3 / 4 contexts), it only comes up beside
    // the others' launches
    std::map<int, int> on_device;
    for (int d : gpu_devices_) on_device[d] = 1;
    auto worth_another_context = [&](int device) -> bool {
        std::lock_guard<std::mutex> l(mu);
        if (!fatal.empty()) return false;
        if (!all_contexts && on_device[device] >= 3) return false;
        long long left = -1;                             // frames no launch has taken yet; -1: unknown
        if (eof) {
            // (a launch holds at least one job: fewer jobs left than contexts already there -- the one file of a one-line list,
            //  whatever its length -- and a further context could never get one)
            const long long jobs_left = (long long)win.size() - (next_launch - base);
            if (!all_contexts && jobs_left <= (long long)(n_first_contexts + contexts_committed.load())) return false;
            left = 0;
            for (long long q = next_launch; q - base < (long long)win.size(); q++) {
                const Item *it = win[(size_t)(q - base)].get();
                if (it->slot.state == 0) { left = -1; break; }      // (a job still in stage 1: its length is not known yet)
                left += it->job.frames;
            }
        }
        const bool yes = all_contexts || left < 0 || left > 1500000LL * (n_first_contexts + contexts_committed.load());
        if (yes) { contexts_committed++; on_device[device]++; }
        return yes;
    };
    auto worker = [&](int g) {
        const int device = gpu_devices_[(size_t)(g % n_log)];
        // This context first: the model and the HIP start-up for a GPU's first one, a share of its weights for the others,
        // then the run's configuration and the buffers of full-size launches.  The list is running meanwhile -- its files
        // are being staged from its first line on, and the contexts that are up take launches.
        {
            std::string cerr;
            // (the worker moves next to its GPU as soon as the context exists -- the runtime is up then, asking for the device's
            //  bus id costs nothing -- and BEFORE the context's staging buffers are reserved: they are this thread's first-touched
            //  pages, so they land on the GPU's NUMA node)
            const int up_rc = BringUpContext(g, plan, cerr, [&](const char *what) {
                trace(g, what);
                if (!single_file && strcmp(what, "ctx: created") == 0) PinToGpuNode(device);
            }, [&, device] { return worth_another_context(device); });
            if (up_rc < 0) return;                  // left out: the contexts that are up finish the list sooner without it
            if (up_rc == 0) {
                std::lock_guard<std::mutex> l(mu);
                // (a clone whose GPU's first context failed says nothing: that context's own message is the run's)
                if (fatal.empty() && !cerr.empty()) fatal = cerr;
                else if (fatal.empty()) fatal_pending_clone = true;
                cv_work.notify_all(); cv_feed.notify_all(); cv_idle.notify_all();
                return;
            }
            const double up = std::chrono::duration<double>(clock::now() - t0).count();
            if (contexts_up.fetch_add(1) == 0) stats_.first_context_seconds = up;
            {
                std::lock_guard<std::mutex> l(mu);
                stats_.create_seconds = std::max(stats_.create_seconds, up);
            }
        }
        Traps &tr = *gpus_[g];
        // (the HIP runtime is up by now: asking for the devices' bus ids costs nothing -- in front of the first context it
        //  waited for the runtime's start-up)
        if (!single_file && pool_->Size() > 0 && !pool_placed.exchange(true)) {
            cpu_set_t all, one;
            CPU_ZERO(&all);
            bool every = true;
            for (int d : gpu_devices_) {
                if (!GpuNodeCpus(d, &one)) { every = false; break; }
                CPU_OR(&all, &all, &one);
            }
            // (only where those nodes hold at least as many usable CPUs as the pool has threads: the pool is sized from every
            //  socket's CPUs, and on a two-socket host with one GPU and the host front-end -- the CPU-bound stage there -- N
            //  threads confined to N/2 cores beside the spinning GPU workers would halve it)
            if (every && CPU_COUNT(&all) >= pool_->Size()) pool_->SetAffinity(all);
        }
        DeviceSlots *slots = dev_slots.count(device) ? dev_slots[device].get() : nullptr;
        // One launch's stay on the device: from its admission until its posterior kernels are done -- the library says
        // so (lcrc_set_kernel_done_callback) while the launch's decoder kernel, labels and posteriors are still on their
        // way: the next context's kernels start meanwhile.  (Released at the latest when the call has returned.)
        struct SlotHold {
            DeviceSlots *s;
            Traps &t;
            const bool registered;
            SlotHold(DeviceSlots *x, Traps &tr) : s(x), t(tr), registered(x != nullptr)
            {
                if (!s) return;
                s->Acquire();
                t.SetKernelDoneCallback([](void *self) { static_cast<SlotHold *>(self)->Leave(); }, this);
            }
            void Leave() { if (s) { s->Release(); s = nullptr; } }
            ~SlotHold()
            {
                if (registered) t.SetKernelDoneCallback(nullptr, nullptr);
                Leave();
            }
        };
        if (!single_file) PinToGpuNode(device);
        std::vector<Item *> items;
        std::vector<int> off;
        auto abort_run = [&](const std::string &msg) {
            std::lock_guard<std::mutex> l(mu);
            if (fatal.empty()) fatal = msg + "\n";
            cv_work.notify_all(); cv_feed.notify_all(); cv_idle.notify_all();
        };
        // labels of a launch the device decoded -> label text of its jobs.  With the decoder overlapped (lists) a launch's
        // labels are fetched after the NEXT staged call has returned (`prev`), or after the last one.
        auto device_labels = [&](const std::vector<Item *> &its, bool prev) -> bool {
            const int cnt = (int)its.size();
            const lcrc_label *lab; const int *lfirst, *lcount; int nu = 0;
            if (!(prev ? tr.PrevLabels(&lab, &lfirst, &lcount, &nu) : tr.LastLabels(&lab, &lfirst, &lcount, &nu)) || nu != cnt) return false;
            pool_->ParallelFor(cnt, [&](int k) {
                CpuTimer tm(stage3_us);
                std::vector<Label> v((size_t)lcount[k]);
                for (int i = 0; i < lcount[k]; i++) {
                    const lcrc_label &lb = lab[lfirst[k] + i];
                    v[i] = Label{lb.start, lb.end, phn_names[lb.phn], lb.score};
                }
                EmitLabels(its[k]->job, mlf != nullptr, v);
            }, 16);
            return true;
        };
        // the launch whose decoder is still running beside the next launch's kernels (lcrc_set_decoder_overlap)
        std::vector<Item *> pending;
        auto finish_pending = [&](bool prev) -> bool {
            if (pending.empty()) return true;
            if (!device_labels(pending, prev)) { abort_run("device decoder returned no labels"); return false; }
            trace(g, "decoded");
            std::lock_guard<std::mutex> l(mu);
            for (Item *it : pending) it->slot.state = 3;
            pending.clear();
            drain();
            return true;
        };
        // A file longer than one launch (-b frames; take_launch hands it over alone).  Its posteriors are computed as
        // consecutive row ranges of at most -b frames, each with the window's reach (15 frames) of context on either side
        // -- a frame's posteriors depend on its 31-frame window only, which is all ProcessOffline's prime / main / flush
        // calls establish (srec.cpp:1035-1059) --, and consumed in order: the decoder takes a range frame by frame, a dump
        // appends it.  Pinned memory and the device buffers stay those of ONE launch whatever the file's length (a 10-hour
        // file as one launch pinned 2.7 GB of posteriors); the features of the whole file live in pageable memory, as the
        // reference's do (srec.cpp:1384-1422 reads whole files).  Same bytes out as the one-launch form (tested).  With -F /
        // -E the file was only stat()ed so far: its features come from the host front-end, i.e. the -E road's bits.
        auto long_job = [&](Item *it) -> bool {
            Job &j = it->job;
            if (j.mel.empty() && in == dfWaveform) {
                CpuTimer tm(stage1_us);
                Stage1(in, out, j, true);
            }
            if (!j.ok) {                                  // unreadable by now: the list stops at this file
                std::lock_guard<std::mutex> l(mu);
                if (stop_seq < 0 || it->seq < stop_seq) stop_seq = it->seq;
                it->slot.state = 3;
                cv_work.notify_all(); cv_feed.notify_all();
                drain();
                return true;
            }
            const int N = j.frames, B = batch_frames_, sh = tr.GetTrapShift();
            std::vector<float> post((size_t)std::min(N, B) * n_out_);
            PhnDec dec;
            FILE *dump = nullptr;
            if (out == dfStrings) {
                dec.SetPhonemes(phn_names_);
                dec.SetStatesPerPhn(states_per_phn_);
                dec.SetTimePruning(time_pruning_);
                dec.SetWPenalty(wpenalty_);
                dec.Init();
                if (n_out_ < dec.NumPhonemes() * states_per_phn_) {
                    j.ok = false;
                    j.err = "posterior vectors are shorter than the phoneme list needs\n";
                }
            } else if (!(dump = BeginHTKRaw(j.tgt, N, n_out_))) {
                j.ok = false;
                j.err = "Can not create file: " + j.tgt + "\n";
            }
            for (int r0 = 0; r0 < N && j.ok; r0 += B) {
                const int rows = std::min(B, N - r0), s0 = std::max(0, r0 - sh), s1 = std::min(N, r0 + rows + sh);
                {
                    SlotHold hold(slots, tr);
                    if (!tr.CalcRows(j.mel.data() + (size_t)s0 * nbanks_, s1 - s0, r0 - s0, rows, post.data())) {
                        if (dump) fclose(dump);
                        abort_run(tr.LastError());
                        return false;
                    }
                }
                kms[g] += tr.LastKernelMs();
                trace(g, "rows done", r0, rows);
                CpuTimer tm(stage3_us);
                if (dump) {
                    if (!AppendHTKRaw(dump, post.data(), rows, n_out_)) { j.ok = false; j.err = "Can not create file: " + j.tgt + "\n"; }
                } else {
                    const long long v0 = ThreadCpuNs();
                    for (int r = 0; r < rows; r++) dec.ProcessFrame(post.data() + (size_t)r * n_out_);
                    viterbi_ns_ += ThreadCpuNs() - v0;
                }
            }
            if (dump && fclose(dump) != 0 && j.ok) { j.ok = false; j.err = "Can not create file: " + j.tgt + "\n"; }
            if (!dump && j.ok) {
                CpuTimer tm(stage3_us);
                dec.Done();
                EmitLabels(j, mlf != nullptr, dec.Labels());
            }
            std::vector<float>().swap(j.mel);
            j.cols = n_out_;
            std::lock_guard<std::mutex> l(mu);
            it->slot.state = 3;
            drain();
            return true;
        };
        const bool row_ranges = tr.HasRowRanges();
        trace(g, "worker up");
        while (take_launch(items)) {
            int cnt = (int)items.size();
            if (cnt == 1 && row_ranges && items[0]->job.frames > batch_frames_) {
                if (!finish_pending(false) || !long_job(items[0])) return;
                continue;
            }
            bool ran_staged = false;             // a staged call of this launch has switched the context's decoder sets
            off.assign(1, 0);
            for (int k = 0; k < cnt; k++) off.push_back(off.back() + items[k]->job.frames);
            trace(g, "took launch", cnt, off.back());
            const auto l0 = clock::now();
            const float *h_post = nullptr;
            std::vector<int> foff;
            if ((FrontendOn() || EnergiesOn()) && in == dfWaveform) {
                // raw bytes in, posteriors out: decode, mel-bank front-end, sentence norm and the three nets all run
                // on the device; the files go straight into the context's pinned byte buffer (read in parallel)
                std::vector<long long> bstart(cnt), blen(cnt);
                long long pos = 0;
                for (int k = 0; k < cnt; k++) {
                    bstart[k] = pos;
                    blen[k] = items[k]->job.file_bytes;
                    pos += blen[k] + (blen[k] & 1);
                }
                unsigned char *pinned = nullptr;
                if (!tr.WaveStageBuffer(pos, &pinned)) { abort_run(tr.LastError()); return; }
                trace(g, "byte buffer", pos);
                std::atomic<int> bad(cnt);     // lowest index whose read failed
                pool_->ParallelFor(cnt, [&](int k) {
                    CpuTimer tm(read_us);
                    if (!ReadWholeFile(items[k]->job.src.c_str(), pinned + bstart[k], blen[k])) {
                        int e = bad.load();
                        while (k < e && !bad.compare_exchange_weak(e, k)) {}
                    }
                }, (int)std::max<long long>(1, (1LL << 20) * cnt / std::max<long long>(1, pos)));
                trace(g, "files read");
                if (bad < cnt) {
                    // The file was there for stage 1's stat() and cannot be read now.  Same meaning as a stage-1
                    // failure (srec.cpp:1246-1290 works file by file): the list stops AT this file -- the jobs in
                    // front of it, in this launch and in the others' launches, are computed and written, then the
                    // error is reported; nothing behind it is written.
                    const int b = bad;
                    {
                        std::lock_guard<std::mutex> l(mu);
                        Job &j = items[b]->job;
                        j.ok = false;
                        j.err = "Can not open waveform file: " + j.src + "\n";
                        if (stop_seq < 0 || items[b]->seq < stop_seq) stop_seq = items[b]->seq;
                        for (int k = b; k < cnt; k++) items[k]->slot.state = 3;
                        cv_work.notify_all(); cv_feed.notify_all();
                        if (b == 0) drain();
                    }
                    if (b == 0) continue;
                    items.resize((size_t)b);
                    cnt = b;
                    off.resize((size_t)b + 1);
                }
                foff.resize(cnt + 1);          // posteriors stay in the context's pinned output buffer
                if (EnergiesOn()) {
                    // -E: the GPU stops at the mel-bank energies (bit for bit the host front-end's); ln() with this host's
                    // libm and the normalisations follow here, in the pinned buffer the posterior kernel then reads in place
                    float *feat = nullptr;
                    if (!tr.WaveStageEnergies(bstart.data(), blen.data(), cnt, &feat, foff.data())) { abort_run(tr.LastError()); return; }
                    if (foff[cnt] != off.back()) { abort_run("frame counts of the launch plan and the GPU front-end differ"); return; }
                    trace(g, "energies back");
                    if (foff[cnt] > 0) {
                        const float shift = C.GetFloat("framenorm", "shift"), floor_ = C.GetFloat("framenorm", "min_floor");
                        pool_->ParallelFor(cnt, [&](int k) {
                            CpuTimer tm(stage1_us);
                            float *x = feat + (size_t)foff[k] * nbanks_;
                            const int fr = foff[k + 1] - foff[k];
                            const size_t nv = (size_t)fr * nbanks_;
                            LnInPlace(x, nv);                                                            // sLn, dspc.h:155-160 (veclog.cpp)
                            if (shift != 0.0f) for (size_t i = 0; i < nv; i++) x[i] += shift;            // srec.cpp:1594-1620
                            if (floor_ != -9999.9f) for (size_t i = 0; i < nv; i++) if (x[i] < floor_) x[i] = floor_;
                            if (fr > 0 && sent_mean_norm_) SentenceMeanNorm(x, fr, nbanks_);
                            if (fr > 0 && (sent_max_norm_ || sent_chmax_norm_)) SentenceMaxNorm(x, fr, nbanks_, sent_max_norm_);
                        }, frame_grain(cnt, foff[cnt]));
                        float *h_mel = nullptr, *hp = nullptr;
                        if (!tr.StageBuffers(foff[cnt], &h_mel, &hp) || h_mel != feat) { abort_run("staging buffers moved under -E"); return; }
                        trace(g, "features ready");
                        SlotHold hold(slots, tr);
                        trace(g, "slot");
                        if (!tr.StageRun(foff.data(), cnt)) { abort_run(tr.LastError()); return; }
                        ran_staged = true;
                        trace(g, "run returned");
                        h_post = hp;
                    }
                } else {
                    SlotHold hold(slots, tr);
                    trace(g, "slot");
                    if (!tr.WaveStageRun(bstart.data(), blen.data(), cnt, nullptr, foff.data())) { abort_run(tr.LastError()); return; }
                    // (the launches were planned with FrontendFramesOf(): the library counts the frames of a file the same way)
                    if (foff[cnt] != off.back()) { abort_run("frame counts of the launch plan and the GPU front-end differ"); return; }
                    ran_staged = true;
                    trace(g, "run returned");
                    h_post = tr.StagedPosteriors();
                }
            } else {
                float *h_mel = nullptr, *hp = nullptr;
                if (!tr.StageBuffers(off.back(), &h_mel, &hp)) { abort_run(tr.LastError()); return; }
                pool_->ParallelFor(cnt, [&](int k) {
                    CpuTimer tm(gather_us);
                    Job &j = items[k]->job;
                    memcpy(h_mel + (size_t)off[k] * nbanks_, j.mel.data(), j.mel.size() * sizeof(float));
                    std::vector<float>().swap(j.mel);
                }, frame_grain(cnt, 2LL * off.back()));
                {
                    SlotHold hold(slots, tr);
                    if (!tr.StageRun(off.data(), cnt)) { abort_run(tr.LastError()); return; }
                    ran_staged = true;
                }
                foff = off;
                h_post = hp;
            }
            if (first_launch.exchange(false))
                stats_.first_launch_seconds = std::chrono::duration<double>(clock::now() - l0).count();
            if (off.back() > 0) kms[g] += tr.LastKernelMs();
            if (dev_dec && dec_overlap) {
                // the launch before this one: its decoder ran beside this launch's kernels
                if (!finish_pending(ran_staged)) return;
                if (off.back() > 0) { pending = items; continue; }       // (this one's labels: after the next launch, or at the end)
                pool_->ParallelFor(cnt, [&](int k) { EmitLabels(items[k]->job, mlf != nullptr, {}); });
            } else if (dev_dec) {
                if (off.back() > 0 && !device_labels(items, false)) { abort_run("device decoder returned no labels"); return; }
                if (off.back() == 0) pool_->ParallelFor(cnt, [&](int k) { EmitLabels(items[k]->job, mlf != nullptr, {}); });
            } else {
                pool_->ParallelFor(cnt, [&](int k) {
                    CpuTimer tm(stage3_us);
                    Job &j = items[k]->job;
                    j.cols = n_out_;
                    Stage3(out, j, mlf != nullptr, const_cast<float *>(h_post) + (size_t)foff[k] * n_out_, n_out_, true);
                }, frame_grain(cnt, off.back()));
            }
            trace(g, "decoded");
            std::lock_guard<std::mutex> l(mu);
            for (Item *it : items) it->slot.state = 3;
            drain();
        }
        if (!finish_pending(false)) return;      // the last launch's labels
        trace(g, "worker done");
    };

    std::vector<std::thread> workers;
    for (int g = 0; g < n_ctx; g++) workers.emplace_back(worker, g);
    if (!need_gpu) for (int k = 0; k < 2; k++) workers.emplace_back(host_worker);

    // ---- feeder ----
    // Jobs whose stage 1 is next to nothing (-F: a stat(); the file is read when its launch is assembled) go to the pool
    // in chunks; the others one by one (a file's read + front-end is a task worth a thread by itself).
    // (the others too once their measured stage 1 turns out short -- lists of very short files: a chunk is sized to ~200 us)
    const bool cheap_stage1 = need_gpu && (FrontendOn() || EnergiesOn()) && in == dfWaveform;
    auto chunk_max = [&]() -> size_t {
        if (single_file) return 1;
        if (cheap_stage1) return 32;
        const long long jobs = stage1_jobs.load();
        if (jobs < 64) return 1;
        const long long avg_ns = stage1_us.load() / jobs;
        return (size_t)std::max<long long>(1, std::min<long long>(32, 200000 / std::max<long long>(1, avg_ns)));
    };
    const int max_pending_jobs = max_pending * 32;
    std::vector<Item *> chunk;
    auto flush_chunk = [&]() {
        if (chunk.empty()) return;
        auto its = std::make_shared<std::vector<Item *>>(std::move(chunk));
        chunk.clear();
        pool_->Submit([&stage1_task, its] { stage1_task(*its); });
    };
    std::string parse_err;
    for (;;) {
        {
            std::unique_lock<std::mutex> l(mu);
            for (;;) {
                if (!fatal.empty() || stop_seq >= 0) break;
                // (until the first context is up nothing takes launches: the host front-end works ahead meanwhile -- 0.1-0.2 s
                //  of every core -- within a byte budget instead of the few launches' worth of frames of the steady state)
                const bool frames_ok = contexts_up.load() == 0 && need_gpu ? staged_bytes < start_up_bytes : staged_frames < max_staged_frames;
                const bool room = win.size() < max_window && frames_ok && staged_bytes < max_staged_bytes;
                if (room && pending1 < max_pending_jobs) break;
                if (!chunk.empty()) {                  // never wait on jobs that have not been handed to the pool yet
                    l.unlock();
                    flush_chunk();
                    l.lock();
                    continue;
                }
                if (!room && pending1 == 0 && !feeder_blocked) {      // what is staged now is all a launch can get
                    feeder_blocked = true;
                    cv_work.notify_all();
                }
                cv_feed.wait(l);
            }
            feeder_blocked = false;
            if (!fatal.empty() || stop_seq >= 0) break;
        }
        std::unique_ptr<Item> it(new Item);
        const int r = next(it->job);
        if (r < 0) parse_err = err_;
        if (r <= 0) break;
        if (verbose_) Log(it->job.tgt.empty() ? it->job.src + "\n" : it->job.src + " -> " + it->job.tgt + "\n");
        Item *raw = it.get();
        {
            std::lock_guard<std::mutex> l(mu);
            raw->seq = base + (long long)win.size();
            win.push_back(std::move(it));
            pending1++;
        }
        chunk.push_back(raw);
        if (chunk.size() >= chunk_max()) flush_chunk();
    }
    flush_chunk();                                     // (also behind an error: every job in the window gets its stage 1)
    {
        std::unique_lock<std::mutex> l(mu);
        eof = true;
        cv_work.notify_all();
    }
    for (auto &t : workers) t.join();
    {
        // every queued stage-1 task must have ended before the window goes out of scope
        std::unique_lock<std::mutex> l(mu);
        cv_idle.wait(l, [&] { return pending1 == 0; });
    }
    if (trace_on) {
        std::sort(trace_rows.begin(), trace_rows.end());
        for (const std::string &r : trace_rows) fputs(r.c_str(), stderr);
    }
    stats_.stage1_seconds += stage1_us.load() * 1e-9 / std::max(1, pool_->Size());
    stats_.cpu_stage1 += stage1_us.load() * 1e-9;
    stats_.cpu_read += read_us.load() * 1e-9;
    stats_.cpu_gather += gather_us.load() * 1e-9;
    stats_.cpu_stage3 += stage3_us.load() * 1e-9;
    stats_.cpu_viterbi = viterbi_ns_.load() * 1e-9;
    stats_.host_threads = std::max(1, pool_->Size());
    for (double k : kms) stats_.gpu_kernel_ms += k;
    stats_.contexts = contexts_up.load();
    if (verbose_ && need_gpu && !single_file) {
        char line[160];
        snprintf(line, sizeof line, "Device path: %d of %d planned contexts came up and took launches\n", contexts_up.load(), n_ctx);
        Log(line);
    }
    stats_.files += files_done;
    stats_.frames += frames_done;
    stats_.seconds += std::chrono::duration<double>(clock::now() - t1).count();
    if (!fatal.empty()) return Fail(fatal);
    if (fatal_pending_clone) return Fail("a GPU context could not be created\n");
    // a job whose stage 1 failed: everything before it has been written
    for (auto &it : win)
        if (!it->job.ok) return Fail(it->job.err);
    if (!parse_err.empty()) return Fail(parse_err);
    return true;
}

bool SpeechRec::ProcessFileListLine(DataFormat in, DataFormat out, const std::string &line)
{
    bool given = false;
    return RunPipeline(in, out, [&](Job &j) -> int {
        if (given) return 0;
        given = true;
        return ParseLine(line, out, false, j) ? 1 : -1;
    }, nullptr, true);
}

bool SpeechRec::ProcessFileList(DataFormat in, DataFormat out, const std::string &list, const std::string &mlf_path)
{
    FILE *fl = fopen(list.c_str(), "r");
    if (!fl) return Fail("Can not open the file list: " + list + "\n");
    {
        struct stat st;
        list_bytes_ = fstat(fileno(fl), &st) == 0 ? (long long)st.st_size : 0;
        long_list_ = list_bytes_ >= 4096;
    }
    FILE *mlf = nullptr;
    if (!mlf_path.empty()) {
        mlf = fopen(mlf_path.c_str(), "w");
        if (!mlf) { fclose(fl); return Fail("Can not create the MLF: " + mlf_path + "\n"); }
        setvbuf(mlf, nullptr, _IOFBF, 1 << 20);      // entries leave under the pipeline's lock: a write() per MB, not per 4 KB
        fprintf(mlf, "#!MLF!#\n");
    }
    char buf[1024];
    // an invalid line stops the list there; the lines before it are processed first (srec.cpp:1246-1290 works line by line)
    const bool ok = RunPipeline(in, out, [&](Job &j) -> int {
        if (!fgets(buf, 1023, fl)) return 0;
        return ParseLine(buf, out, mlf != nullptr, j) ? 1 : -1;
    }, mlf, false);
    if (mlf) fclose(mlf);
    fclose(fl);
    return ok;
}

}  // namespace phnrec
