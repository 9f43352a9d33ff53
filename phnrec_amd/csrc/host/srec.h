// srec.h -- orchestration of the drop-in `phnrec` CLI: the part of the reference's
// SpeechRec (srec.cpp) that surrounds the posterior path -- configuration, file / list
// modes, the wf -> par -> post -> str data-format ladder, HTK dumps, label and MLF output.
// The par -> post step is the GPU path (class Traps over include/lcrc.h); everything else
// here is host plumbing kept byte-compatible with the reference's files and messages.
//
// What is different by design: utterances are independent, so a file list is processed in
// chunks -- front-end and Viterbi on a host thread pool, posteriors in multi-utterance
// launches spread over all selected GPUs (no exchange between GPUs; outputs are written in
// list order).  Live audio (-a) and the STK decoder are outside this path's scope.
#ifndef PHNREC_HOST_SREC_H
#define PHNREC_HOST_SREC_H

#include <cstdio>
#include <condition_variable>
#include <deque>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
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
    double seconds = 0, gpu_kernel_ms = 0;      // seconds: processing without one-off GPU/pool set-up
    double init_seconds = 0, stage1_seconds = 0;
};

// Persistent worker pool; ParallelFor may be called from several threads at once (each GPU
// worker farms its gather / decode loops out to the same pool).
int UsableCpus();   // affinity mask capped by the cgroup CPU quota

class ThreadPool {
public:
    explicit ThreadPool(int n);
    ~ThreadPool();
    void ParallelFor(int n, const std::function<void(int)> &fn);
    int Size() const { return (int)threads_.size(); }

private:
    struct Task { const std::function<void(int)> *fn; int begin, end; struct Group *group; };
    void Run();
    std::vector<std::thread> threads_;
    std::deque<Task> queue_;
    std::mutex mu_;
    std::condition_variable cv_;
    bool stop_ = false;
};

class SpeechRec {
public:
    bool Init(const std::string &config_file);             // srec.cpp:235-707
    void SetVerbose(bool v) { verbose_ = v; }
    void SetWaveFormat(WaveFormat f) { wave_.format = f; }
    void SetWPenalty(float p) { wpenalty_ = p; wpenalty_set_ = true; }
    void SetGpus(int n) { n_gpus_ = n; }
    void SetBatchFrames(int n) { batch_frames_ = n; }
    void SetHostThreads(int n) { host_threads_ = n; }
    void SetGpuDecoder(bool v) { gpu_decoder_ = v; }     // -D: PhnDec on the GPU, posteriors never leave it
    void SetSplitF16(bool v) { split_f16_ = v; }         // -H: lcrc_set_arithmetic(LCRC_ARITH_SPLIT_F16)
    void SetGpuFrontend(bool v) { gpu_frontend_ = v; }   // -F: waveform -> posteriors without the host front-end
    // srec.cpp:1201-1244: "src[ \t]+tgt" or "src" (target derived)
    bool ProcessFileListLine(DataFormat in, DataFormat out, const std::string &line);
    bool ProcessFileList(DataFormat in, DataFormat out, const std::string &list, const std::string &mlf);
    const std::string &LastError() const { return err_; }
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
    bool RunJobs(DataFormat in, DataFormat out, std::vector<Job> &jobs, FILE *mlf);
    void Stage1(DataFormat in, DataFormat out, Job &job);              // load [+ front-end] [+ sentence norm]
    // soft funcs, decode / dump; `post` = job.frames x cols posteriors (writable)
    // device_done: softening (and, for dumps, the big-endian byte order) already applied by the GPU
    void Stage3(DataFormat out, Job &job, bool mlf, float *post, int cols, bool device_done = false);
    void EmitLabels(Job &job, bool mlf, const std::vector<Label> &labels);
    bool EnsureGpus();
    void Log(const std::string &msg) const { if (verbose_) fputs(msg.c_str(), stdout); }
    bool Fail(const std::string &msg) { err_ = msg; return false; }
    std::string LabelNameForMlf(const std::string &file) const;        // srec.cpp:1424-1436

    std::string config_dir_, err_;
    bool verbose_ = false, traps_enabled_ = true, sent_mean_norm_ = false, gpu_frontend_ = false, gpu_decoder_ = false, split_f16_ = false;
    WaveOptions wave_;
    int nbanks_ = 15, n_out_ = 0, n_gpus_ = 0, batch_frames_ = 32768, host_threads_ = 0;
    float wpenalty_ = -2.0f;
    bool wpenalty_set_ = false;
    int states_per_phn_ = 1, time_pruning_ = 40;
    std::string post_soft_ = "none", dec_soft_ = "log";
    float post_soft_arg_[3] = {0, 0, 0}, dec_soft_arg_[3] = {0, 0, 0};
    std::vector<std::string> phonemes_path_;
    std::string phoneme_list_;
    std::vector<std::unique_ptr<Traps>> gpus_;        // two contexts per GPU (alternating launches)
    std::unique_ptr<ThreadPool> pool_;
    RunStats stats_;
    MelBanks mb_proto_;
};

// file-name helpers with the reference's semantics (filename.cpp:30-46,100-128)
std::string ChangeFileSuffix(const std::string &name, const std::string &suffix);
std::string ChangeFilePath(const std::string &name, const std::string &new_path);
std::string GetFilePath(const std::string &name);

}  // namespace phnrec
#endif
