// srec.h -- orchestration of the drop-in `phnrec` CLI: the part of the reference's
// SpeechRec (srec.cpp) that surrounds the posterior path -- configuration, file / list
// modes, the wf -> par -> post -> str data-format ladder, HTK dumps, label and MLF output.
// The par -> post step is the GPU path (class Traps over include/lcrc.h); everything else
// here is host plumbing kept byte-compatible with the reference's files and messages.
//
// What is different by design: utterances are independent, so a file list runs as ONE pipeline
// from its first line to its last -- read-ahead / front-end on a host thread pool, posteriors in
// multi-utterance launches that all GPU contexts pull from one queue (no exchange between GPUs),
// Viterbi / dump writing behind each launch, outputs in list order.  Nothing joins the GPUs
// before the end of the list.  Live audio (-a) and the STK decoder are outside this path's scope.
#ifndef PHNREC_HOST_SREC_H
#define PHNREC_HOST_SREC_H

#include <cstdio>
#include <condition_variable>
#include <deque>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include <sched.h>

#include <atomic>
#include <string>
#include <vector>

#include "config.h"
#include "frontend.h"
#include "phndec.h"
#include "traps.h"

namespace phnrec {

enum DataFormat { dfUnknown = 0, dfWaveform, dfParams, dfPosteriors, dfStrings };
DataFormat ParseDataFormat(const std::string &s);          // wf | par | post | str

struct RunStats {
    long long frames = 0, files = 0;
    double seconds = 0, gpu_kernel_ms = 0;      // seconds: the list from its first line to its last, context start-up included
    double init_seconds = 0, stage1_seconds = 0;
    // The GPU contexts come up BESIDE the list (each context's worker creates it, then joins the running pipeline):
    // create_seconds = from the start of the run until the last one was up, first_context_seconds = until the first was
    // (HIP start-up, model load, pack, upload: no launch can start sooner) -- both overlap `seconds`, not init_seconds
    double create_seconds = 0, first_context_seconds = 0;
    int contexts = 0;                           // GPU contexts that came up and took launches (of the 2-3 per GPU planned)
    double first_launch_seconds = 0;            // wall time of the run's first launch call (code-object load, cold clock)
    // CPU seconds of the host stages, summed over threads: stage 1 (file read [+ front-end]; stat() with -F), file reads
    // into pinned memory (-F), gather into pinned memory, stage 3 (Viterbi / label formatting / dump writing)
    double cpu_stage1 = 0, cpu_read = 0, cpu_gather = 0, cpu_stage3 = 0;
    double cpu_viterbi = 0;             // of cpu_stage3: the Viterbi frames alone (the rest is label text and output)
    int host_threads = 1;
};

// Persistent worker pool.  ParallelFor (blocking, data-parallel) may be called from several threads at
// once: each GPU worker farms its gather / decode loops out to the same pool, and those chunks run
// BEFORE anything queued with Submit (asynchronous read-ahead tasks), so a launch that is ready never
// waits behind the front-end of files that are not needed yet.
int UsableCpus();   // affinity mask capped by the cgroup CPU quota

// Host thread pool.  Two things keep its CPU seconds down on hosts where a core that has slept runs slowly for a
// while (cold clock, cold caches: the per-frame CPU time of the host Viterbi was 2.7 x higher with sixteen
// intermittently busy threads than with one busy one, gpurun_out/r04_host_decoder_probe.txt):
//   * idle threads wait on a STACK -- work wakes the thread that went idle last, so a load that needs five cores keeps
//     five threads busy and warm instead of touring all sixteen;
//   * ParallelFor takes a grain (items per chunk) and the CALLER works through chunks too instead of sleeping.
class ThreadPool {
public:
    explicit ThreadPool(int n);
    ~ThreadPool();
    // fn(i) for i in [0, n); chunks of at least `grain` items; returns when all have run (the caller runs chunks too)
    void ParallelFor(int n, const std::function<void(int)> &fn, int grain = 1);
    void Submit(std::function<void()> fn);                  // runs inline when the pool has no threads
    int Size() const { return (int)threads_.size(); }
    // CPUs the pool's threads keep to from now on (each thread applies it before its next task); empty set: no change
    void SetAffinity(const cpu_set_t &set);

private:
    struct Group { int pending = 0; std::condition_variable cv; };
    struct Task { const std::function<void(int)> *fn; int begin, end; Group *group; };
    struct Worker { std::condition_variable cv; bool wake = false; int affinity_seen = 0; };
    void Run(int id);
    void WakeLocked(int k);                          // the k most recently idled threads
    void FinishLocked(const Task &t);
    std::vector<std::thread> threads_;
    std::vector<std::unique_ptr<Worker>> workers_;
    std::vector<int> idle_;                          // stack: back() went idle last
    std::deque<Task> queue_;                         // ParallelFor chunks: served first
    std::deque<std::function<void()>> background_;   // Submit tasks
    std::mutex mu_;
    bool stop_ = false;
    cpu_set_t affinity_;
    int affinity_gen_ = 0;
};

class SpeechRec {
public:
    bool Init(const std::string &config_file);             // srec.cpp:235-707
    void SetVerbose(bool v) { verbose_ = v; }
    void SetWaveFormat(WaveFormat f) { wave_.format = f; }
    void SetWPenalty(float p) { wpenalty_ = p; wpenalty_set_ = true; }
    void SetGpus(int n) { n_gpus_ = n; }
    void SetBatchFrames(int n) { batch_frames_ = n; batch_given_ = true; }
    void SetHostThreads(int n) { host_threads_ = n; }
    void SetGpuDecoder(bool v) { gpu_decoder_ = v; }     // -D: PhnDec on the GPU, posteriors never leave it
    void SetSplitF16(bool v) { split_f16_ = v; }         // -H: lcrc_set_arithmetic(LCRC_ARITH_SPLIT_F16)
    void SetGpuFrontend(bool v) { gpu_frontend_ = v; }   // -F: waveform -> posteriors without the host front-end
    // -E: the front-end's FFTs and bank sums on the GPU, ln() and the normalisations on the host: the host front-end's
    // features bit for bit at a tenth of its CPU time
    void SetGpuEnergies(bool v) { gpu_energies_ = v; }
    // Starts the HIP runtime and the primary context of EVERY distinct device the run will use (SetGpus first: -g N,
    // PHNREC_DEVICE_MAP) on helper threads, one per device (lcrc_device_warmup): ~0.1-0.2 s each that then overlap with
    // Init(), the model files, their re-packing and each other.  Call it as early as the conversion is known to need the GPU.
    // frontend / decoder: the run will (or may) use the GPU front-end / the device decoder -- a list, or the flags given --:
    // their code objects are brought onto the devices too (lcrc_device_preload)
    void WarmUpGpuAsync(bool frontend = false, bool decoder = false);
    void JoinWarmUp() { for (auto &t : warmup_) if (t.joinable()) t.join(); }
    ~SpeechRec();
    // srec.cpp:1201-1244: "src[ \t]+tgt" or "src" (target derived)
    bool ProcessFileListLine(DataFormat in, DataFormat out, const std::string &line);
    bool ProcessFileList(DataFormat in, DataFormat out, const std::string &list, const std::string &mlf);
    const std::string &LastError() const { return err_; }
    // the device path the last list took: "host" | "E" | "F", "+D" with the decoder on the GPU, ",auto" behind what
    // RunPipeline chose by itself (PHNREC_STATS prints it)
    std::string ModeString() const
    {
        std::string m = FrontendOn() ? "F" : EnergiesOn() ? "E" : "host";
        if (DecoderOn()) m += "+D";
        if (auto_frontend_ || auto_energies_ || auto_decoder_) m += ",auto";
        return m;
    }
    const RunStats &Stats() const { return stats_; }
    Config C;

private:
    struct Job {
        std::string src, tgt;
        std::vector<float> mel;            // [frames][nbanks] (par) or posteriors when in == post
        std::vector<float> post;
        std::vector<unsigned char> bytes;  // raw file (GPU front-end mode)
        int frames = 0, cols = 0;
        long long file_bytes = 0;          // -F: size of the waveform file (read later, into pinned memory)
        std::string labels;                // formatted label / MLF text
        bool ok = true;
        std::string err;
    };
    bool ParseLine(const std::string &line, DataFormat out, bool mlf, Job &job);
    // next(job): 1 = a job, 0 = end of the list, -1 = invalid line (LastError() says which)
    bool RunPipeline(DataFormat in, DataFormat out, const std::function<int(Job &)> &next, FILE *mlf, bool single_file);
    // load [+ front-end] [+ sentence norm]; host_features: the host front-end even where -F / -E would only stat() the file
    void Stage1(DataFormat in, DataFormat out, Job &job, bool host_features = false);
    // soft funcs, decode / dump; `post` = job.frames x cols posteriors (writable)
    // device_done: softening (and, for dumps, the big-endian byte order) already applied by the GPU
    void Stage3(DataFormat out, Job &job, bool mlf, float *post, int cols, bool device_done = false);
    void EmitLabels(Job &job, bool mlf, const std::vector<Label> &labels);
    // what every context of a run is configured with before it takes its first launch
    struct ContextPlan {
        lcrc_softening soft[2];
        int n_soft = 1;
        bool big_endian = false, device_decoder = false, decoder_overlap = false, launch_order = false;
        int poll_us = 0;
        int reserve_rows = 0;              // 0: buffers grow with the first launches
        long long reserve_wave_bytes = 0;
    };
    bool DeviceMap(std::vector<int> &devices);            // physical device of each of the -g N logical GPUs
    bool PlanGpus(int contexts_per_gpu);                  // device map + one (empty) place per context
    // context `idx` (k * n_gpus + g): created if it does not exist yet -- the first of a GPU loads the model, the others
    // wait for it and share its weights --, then configured per `plan`.  Called by the context's own worker thread.
    // mark(what): the steps' names as they end (PHNREC_TRACE_PIPELINE's time line).  worth_it(): asked once, when a context
    // that shares another's weights could be created (its base is up): false = the list will be over before this context
    // could help -- it is not created (and neither are those that would share ITS place).  1 = up, 0 = failed (err), -1 = left out.
    int BringUpContext(int idx, const ContextPlan &plan, std::string &err, const std::function<void(const char *)> &mark,
                       const std::function<bool()> &worth_it);
    std::string SetUpContext(Traps &t);
    int FrontendFramesOf(long long file_bytes) const;     // frames of a waveform file of that size (lcrc_frontend_frames' rule)
    void Log(const std::string &msg) const { if (verbose_) fputs(msg.c_str(), stdout); }
    bool Fail(const std::string &msg) { err_ = msg; return false; }
    std::string LabelNameForMlf(const std::string &file) const;        // srec.cpp:1424-1436

    std::atomic<long long> viterbi_ns_{0};
    std::string config_dir_, err_;
    bool sent_max_norm_ = false, sent_chmax_norm_ = false;
    bool verbose_ = false, traps_enabled_ = true, sent_mean_norm_ = false, gpu_frontend_ = false, gpu_energies_ = false, gpu_decoder_ = false, split_f16_ = false;
    // what RunPipeline switched on by itself for the contexts it built (-g >= 2: -E, -g >= 4: -D); the members above stay the caller's
    bool auto_frontend_ = false, auto_energies_ = false, auto_decoder_ = false;
    bool FrontendOn() const { return gpu_frontend_ || auto_frontend_; }
    bool EnergiesOn() const { return !FrontendOn() && (gpu_energies_ || auto_energies_); }
    bool DecoderOn() const { return gpu_decoder_ || auto_decoder_; }
    bool GpuFrontendTakesConfig();
    WaveOptions wave_;
    int nbanks_ = 15, n_out_ = 0, n_gpus_ = 0, batch_frames_ = 32768, host_threads_ = 0;
    bool long_list_ = false;             // the list file has >= 4 KB (~100 entries): buffers are reserved ahead of the first launch
    long long list_bytes_ = 0;           // size of the list file (how many contexts per GPU a list is worth)
    bool batch_given_ = false;           // -b: otherwise 32 768 frames per launch, 65 536 with the decoder on the GPU
    float wpenalty_ = -2.0f;
    bool wpenalty_set_ = false;
    int states_per_phn_ = 1, time_pruning_ = 40;
    std::string post_soft_ = "none", dec_soft_ = "log";
    float post_soft_arg_[3] = {0, 0, 0}, dec_soft_arg_[3] = {0, 0, 0};
    std::vector<std::string> phonemes_path_;
    std::string phoneme_list_;
    std::vector<std::string> phn_names_;              // dicts/phoneme_list, read once in Init
    std::vector<std::unique_ptr<Traps>> gpus_;        // [context k of GPU g] at k * n_gpus + g; null until its worker has built it
    std::vector<int> ctx_state_;                      // per context: 0 not yet, 1 up, -1 failed, -2 left out (ctx_mu_)
    std::mutex ctx_mu_;
    std::condition_variable ctx_cv_;
    int fe_vector_size_ = 0, fe_vector_step_ = 1;    // melbanks/vector_size, vector_step (FrontendFramesOf)
    std::vector<int> gpu_devices_;                    // physical device of each logical GPU (PHNREC_DEVICE_MAP)
    std::unique_ptr<ThreadPool> pool_;
    std::vector<std::thread> warmup_;
    RunStats stats_;
    MelBanks mb_proto_;
};

// file-name helpers with the reference's semantics (filename.cpp:30-46,100-128)
std::string ChangeFileSuffix(const std::string &name, const std::string &suffix);
std::string ChangeFilePath(const std::string &name, const std::string &new_path);
std::string GetFilePath(const std::string &name);

}  // namespace phnrec
#endif
