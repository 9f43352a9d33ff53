// srec.cpp -- see srec.h
#include "srec.h"

#include <sched.h>
#include <sys/stat.h>

#include <atomic>
#include <chrono>
#include <cmath>
#include <cstring>
#include <functional>
#include <thread>

#include "htk.h"

namespace phnrec {

// ---- small helpers ---------------------------------------------------------------------

DataFormat ParseDataFormat(const std::string &s)
{
    if (s == "wf") return dfWaveform;
    if (s == "par") return dfParams;
    if (s == "post") return dfPosteriors;
    if (s == "str") return dfStrings;
    return dfUnknown;
}

static size_t LastSep(const std::string &s)
{
    const size_t a = s.rfind('/'), b = s.rfind('\\');
    if (a == std::string::npos) return b;
    if (b == std::string::npos) return a;
    return a > b ? a : b;
}

// Replace what follows the last '.' of the base name, or append ".suffix" (filename.cpp:30-46)
std::string ChangeFileSuffix(const std::string &name, const std::string &suffix)
{
    const size_t dot = name.rfind('.'), sep = LastSep(name);
    if (dot == std::string::npos || (sep != std::string::npos && sep > dot)) return name + "." + suffix;
    return name.substr(0, dot + 1) + suffix;
}

// Replace the directory part; unchanged when the name has no separator (filename.cpp:100-114)
std::string ChangeFilePath(const std::string &name, const std::string &new_path)
{
    const size_t sep = LastSep(name);
    if (sep == std::string::npos) return name;
    return new_path + name.substr(sep);
}

// Strip the last component; "" when there is no separator (filename.cpp:116-128)
std::string GetFilePath(const std::string &name)
{
    const size_t sep = LastSep(name);
    return sep == std::string::npos ? std::string() : name.substr(0, sep);
}

struct Group {
    std::mutex mu;
    std::condition_variable cv;
    int pending = 0;
};

// CPUs this process may actually use: the affinity mask, capped by the cgroup CPU quota (containers
// commonly show all of the host's cores but grant a fraction; a pool sized to the former is throttled).
int UsableCpus()
{
    int n = (int)std::max(1u, std::thread::hardware_concurrency());
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof set, &set) == 0) n = std::min(n, std::max(1, CPU_COUNT(&set)));
    long long quota = -1, period = 0;
    if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {                 // cgroup v2: "<quota|max> <period>"
        char q[64];
        if (fscanf(f, "%63s %lld", q, &period) == 2 && strcmp(q, "max") != 0) quota = atoll(q);
        fclose(f);
    } else if (FILE *f1 = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {   // cgroup v1
        if (fscanf(f1, "%lld", &quota) != 1) quota = -1;
        fclose(f1);
        if (FILE *f2 = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) {
            if (fscanf(f2, "%lld", &period) != 1) period = 0;
            fclose(f2);
        }
    }
    if (quota > 0 && period > 0) n = std::min(n, (int)std::max(1LL, (quota + period - 1) / period));
    return n;
}

ThreadPool::ThreadPool(int n)
{
    for (int i = 0; i < n; i++) threads_.emplace_back([this] { Run(); });
}

ThreadPool::~ThreadPool()
{
    {
        std::lock_guard<std::mutex> l(mu_);
        stop_ = true;
    }
    cv_.notify_all();
    for (auto &t : threads_) t.join();
}

void ThreadPool::Run()
{
    for (;;) {
        Task t;
        {
            std::unique_lock<std::mutex> l(mu_);
            cv_.wait(l, [this] { return stop_ || !queue_.empty(); });
            if (queue_.empty()) return;
            t = queue_.front();
            queue_.pop_front();
        }
        for (int i = t.begin; i < t.end; i++) (*t.fn)(i);
        std::lock_guard<std::mutex> l(t.group->mu);
        if (--t.group->pending == 0) t.group->cv.notify_all();
    }
}

void ThreadPool::ParallelFor(int n, const std::function<void(int)> &fn)
{
    if (n <= 0) return;
    if (threads_.empty() || n == 1) {
        for (int i = 0; i < n; i++) fn(i);
        return;
    }
    // a few chunks per worker keeps the tail short without flooding the queue
    const int chunks = std::min(n, 4 * (int)threads_.size());
    Group g;
    g.pending = chunks;
    {
        std::lock_guard<std::mutex> l(mu_);
        for (int c = 0; c < chunks; c++)
            queue_.push_back(Task{&fn, (int)((long long)n * c / chunks), (int)((long long)n * (c + 1) / chunks), &g});
    }
    cv_.notify_all();
    std::unique_lock<std::mutex> l(g.mu);
    g.cv.wait(l, [&g] { return g.pending == 0; });
}

// posteriors/softening_func and decoder/softening_func (srec.cpp:164-176, srec.h:192-194)
static float Soften(const std::string &f, float v, const float *a)
{
    if (f == "none") return v;
    if (f == "log") return logf(v);
    if (f == "igor") {
        if (v < a[0]) return logf(v * (1.0f / a[0])) / logf(a[2]);
        return -1.0f * logf((1.0f + (-1.0f * v)) * (1.0f / (1.0f - a[0]))) / logf(a[1]);
    }
    return sqrtf(-2.0f * logf(v));                       // gmm_bypass
}

static bool ParseSoftFunc(const std::string &s, std::string &name, float *a)
{
    char func[256];
    a[0] = a[1] = a[2] = 0.0f;
    if (sscanf(s.c_str(), "%255s %f %f %f", func, &a[0], &a[1], &a[2]) != 4) return false;
    name = func;
    return name == "none" || name == "log" || name == "igor" || name == "gmm_bypass";
}

// ---- Init ------------------------------------------------------------------------------

bool SpeechRec::Init(const std::string &config_file)
{
    config_dir_ = GetFilePath(config_file);
    int line = 0;
    char msg[1200];
    switch (C.Load(config_file, &line)) {
    case Config::OK: break;
    case Config::UNKVAR:
        snprintf(msg, sizeof msg, "Unknown variable in configuration file '%s', line %d\n", config_file.c_str(), line);
        return Fail(msg);
    case Config::BADVAL:
        snprintf(msg, sizeof msg, "Invalid argument for a vatiable in configuration file '%s', line %d\n", config_file.c_str(), line);
        return Fail(msg);
    case Config::FILEERR:
        snprintf(msg, sizeof msg, "Can not open configuration file '%s'\n", config_file.c_str());
        return Fail(msg);
    case Config::INVVAR:
        snprintf(msg, sizeof msg, "Invalid notation of variable in configuration file '%s', line %d\n", config_file.c_str(), line);
        return Fail(msg);
    }
    // $C / $T substitution (srec.cpp:268-332); the temp dir is created, failure ignored
    C.SetString("dirs", "tmp", C.Subst(C.GetString("dirs", "tmp"), config_dir_));
    mkdir(C.GetString("dirs", "tmp").c_str(), 0777);
    const char *paths[][2] = {{"models", "hmm_defs"}, {"dicts", "phoneme_list"}, {"networks", "default"},
                              {"dicts", "lexicon1"}, {"dicts", "lexicon2"}, {"dicts", "keyword_list"},
                              {"kws", "thresholds_file"}, {"gptransc", "rules"}, {"gptransc", "symbols"},
                              {"onlinenorm", "file"}};
    for (auto &p : paths) C.SetString(p[0], p[1], C.Subst(C.GetString(p[0], p[1]), config_dir_));

    // source
    wave_.format = ParseWaveFormat(C.GetString("source", "format"));
    if (wave_.format == WF_UNKNOWN) {
        snprintf(msg, sizeof msg, "Invalid waveform format '%s'. Supported data formats are 'lin16' and 'alaw'.\n",
                 C.GetString("source", "format").c_str());
        return Fail(msg);
    }
    wave_.scale = C.GetFloat("source", "scale");
    wave_.dc_shift = C.GetFloat("source", "dc_shift");
    wave_.noise_level = C.GetFloat("source", "noise_level");

    Log("\nSystem initialization\n");
    if (C.GetString("params", "kind") != "fbanks") {
        snprintf(msg, sizeof msg, "Unknown parameterization (parameters/kind): '%s'\n", C.GetString("params", "kind").c_str());
        return Fail(msg);
    }
    Log("  - mel-banks ...\n");
    nbanks_ = C.GetInt("melbanks", "nbanks");
    mb_proto_.Configure(nbanks_, C.GetInt("melbanks", "nbanks_full"), C.GetInt("source", "sample_freq"),
                        C.GetInt("melbanks", "vector_size"), C.GetInt("melbanks", "vector_step"),
                        C.GetFloat("melbanks", "preem_coef"), C.GetBool("melbanks", "z_mean_source"),
                        C.GetFloat("melbanks", "lower_freq"), C.GetFloat("melbanks", "higher_freq"));
    Log("  - online normalization ...\n");
    sent_mean_norm_ = C.GetBool("offlinenorm", "sent_mean_norm");
    if (C.GetBool("offlinenorm", "sent_var_norm"))
        return Fail("offlinenorm/sent_var_norm=true is not supported (the reference aborts on it: srec.cpp:1531 reads a variable that is not in its schema)\n");

    Log("  - posteriors (loading NNs) ...\n");
    const std::string sys = C.GetString("posteriors", "system");
    if (sys != "LCRC" && sys != "3BT" && sys != "1BT" && sys != "1BT_DCT") {
        snprintf(msg, sizeof msg, "Unknown system, check configuration: %s", sys.c_str());
        return Fail(msg);
    }
    traps_enabled_ = C.GetBool("posteriors", "enabled");
    if (traps_enabled_) {
        if (C.GetInt("posteriors", "length") != 31 || (sys == "LCRC" && !C.GetBool("posteriors", "add_c0")))
            return Fail("the GPU path implements posteriors/length=31 (and add_c0=true for system=LCRC)\n");
        // host-only validation of the model directory (the GPU is claimed lazily, when a
        // par -> post conversion is actually requested)
        int dims[9];
        if (sys == "LCRC") {
            if (lcrc_model_info(config_dir_.c_str(), nbanks_, dims, nullptr, 0, nullptr) != LCRC_OK) {
                snprintf(msg, sizeof msg, "%s\n", lcrc_last_error(nullptr));
                return Fail(msg);
            }
            n_out_ = dims[8];
        } else {
            n_out_ = lcrc_model_outputs(config_dir_.c_str(), sys.c_str());
            if (n_out_ < 0) {
                snprintf(msg, sizeof msg, "%s\n", lcrc_last_error(nullptr));
                return Fail(msg);
            }
        }
    }

    Log("  - decoder ...\n\n");
    const std::string dtype = C.GetString("decoder", "type");
    if (dtype != "phndec") {
        snprintf(msg, sizeof msg, "Unknown dekoder, check configuration: %s", dtype.c_str());
        return Fail(msg);
    }
    states_per_phn_ = C.GetInt("decoder", "num_states_per_phn");
    time_pruning_ = C.GetInt("decoder", "time_pruning");
    phoneme_list_ = C.GetString("dicts", "phoneme_list");
    {
        PhnDec probe;
        if (!probe.LoadPhnList(phoneme_list_)) {
            snprintf(msg, sizeof msg, "Can not load phoneme list: %s", phoneme_list_.c_str());
            return Fail(msg);
        }
    }
    if (!wpenalty_set_) wpenalty_ = C.GetFloat("decoder", "wpenalty");
    if (C.GetString("decoder", "mode") == "kws") return Fail("decoder/mode=kws needs the STK decoder, which is outside this path\n");
    if (!ParseSoftFunc(C.GetString("posteriors", "softening_func"), post_soft_, post_soft_arg_) ||
        !ParseSoftFunc(C.GetString("decoder", "softening_func"), dec_soft_, dec_soft_arg_))
        return Fail("Invalid softening function format. The format should be function identificator and three floating point arguments.\n");

    if (verbose_) {
        printf("------------------- SUMMARY -------------------\n");
        printf("Dictionary:   %s\n", C.GetString("dicts", "phoneme_list").c_str());
        printf("Network file: %s\n", C.GetString("networks", "default").c_str());
        printf("HMM file:     %s\n", C.GetString("models", "hmm_defs").c_str());
        printf("#States/Phn:  %d\n", C.GetInt("models", "nstates"));
        printf("Time pruning: %d\n", C.GetInt("decoder", "time_pruning"));
        printf("Word penalty: %f\n", C.GetFloat("decoder", "wpenalty"));
        printf("Soft func:    %s\n", C.GetString("decoder", "softening_func").c_str());
        printf("-----------------------------------------------\n\n");
    }
    return true;
}

bool SpeechRec::EnsureGpus()
{
    if (!gpus_.empty()) return true;
    int n = n_gpus_;
    if (n <= 0) n = 1;
    // several contexts per GPU: while one's launch is in flight the others stage / decode / write
    // (PHNREC_CTX_PER_GPU overrides the default for experiments)
    int per_gpu = 2;
    if (const char *e = getenv("PHNREC_CTX_PER_GPU")) per_gpu = std::max(1, std::min(8, atoi(e)));
    // PHNREC_DEVICE_MAP="0,0": the physical device of each of the -g N logical GPUs (default: 0..N-1).  Lets a
    // 1-GPU box run the -g 2 arrangement (4 contexts, one launch queue); on an 8-GPU node it picks the GPUs.
    std::vector<int> dmap;
    if (const char *e = getenv("PHNREC_DEVICE_MAP")) {
        for (const char *q = e; *q;) {
            char *end = nullptr;
            const long v = strtol(q, &end, 10);
            if (end == q || v < 0) return Fail(std::string("PHNREC_DEVICE_MAP must be a comma-separated list of GPU indices: ") + e + "\n");
            dmap.push_back((int)v);
            q = *end == ',' ? end + 1 : end;
            if (*end && *end != ',') return Fail(std::string("PHNREC_DEVICE_MAP must be a comma-separated list of GPU indices: ") + e + "\n");
        }
        if ((int)dmap.size() < n) return Fail("PHNREC_DEVICE_MAP names fewer devices than -g asks for\n");
    }
    for (int k = 0; k < per_gpu * n; k++) {
        const int d = dmap.empty() ? k / per_gpu : dmap[k / per_gpu];
        std::unique_ptr<Traps> t(new Traps);
        t->SetSystem(C.GetString("posteriors", "system").c_str());
        t->SetTrapLen(C.GetInt("posteriors", "length"));
        t->SetHamming(C.GetBool("posteriors", "hamming"));
        t->SetNBanks(nbanks_);
        t->SetAddC0(C.GetBool("posteriors", "add_c0"));
        t->SetBunchSize(atoi(C.GetString("posteriors", "bunch_size").c_str()));
        t->SetDevice(d);
        // outputs must not depend on how files are packed into launches (-g 1 and -g N write the same bytes)
        t->SetHiddenSplit(1);
        if (!t->Init(config_dir_.c_str())) return Fail(t->LastError() + "\n");
        if (split_f16_ && !t->SetArithmetic(LCRC_ARITH_SPLIT_F16)) return Fail(t->LastError() + "\n");
        if (gpu_frontend_) {
            if (wave_.noise_level != 0.0f) return Fail("source/noise_level needs the host front-end (libc rand()); drop -F\n");
            lcrc_frontend fe;
            fe.wave_format = wave_.format == WF_LIN16 ? 1 : 2;
            fe.sample_freq = C.GetInt("source", "sample_freq");
            fe.vector_size = C.GetInt("melbanks", "vector_size");
            fe.vector_step = C.GetInt("melbanks", "vector_step");
            fe.nbanks_full = C.GetInt("melbanks", "nbanks_full");
            fe.lower_freq = C.GetFloat("melbanks", "lower_freq");
            fe.higher_freq = C.GetFloat("melbanks", "higher_freq");
            fe.preem_coef = C.GetFloat("melbanks", "preem_coef");
            fe.scale = wave_.scale;
            fe.dc_shift = wave_.dc_shift;
            fe.z_mean_source = C.GetBool("melbanks", "z_mean_source") ? 1 : 0;
            fe.sent_mean_norm = sent_mean_norm_ ? 1 : 0;
            if (C.GetFloat("framenorm", "shift") != 0.0f || C.GetFloat("framenorm", "min_floor") != -9999.9f)
                return Fail("framenorm/* needs the host front-end; drop -F\n");
            if (!t->ConfigureFrontend(fe)) return Fail(t->LastError() + "\n");
        }
        gpus_.push_back(std::move(t));
    }
    return true;
}

// ---- per-utterance stages --------------------------------------------------------------

static bool ReadFile(const std::string &path, std::vector<unsigned char> &bytes)
{
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) return false;
    fseek(f, 0, SEEK_END);
    const long len = ftell(f);
    fseek(f, 0, SEEK_SET);
    bytes.resize((size_t)len);
    const bool ok = len == 0 || fread(bytes.data(), 1, (size_t)len, f) == (size_t)len;
    fclose(f);
    return ok;
}

void SpeechRec::Stage1(DataFormat in, DataFormat out, Job &job)
{
    char msg[1200];
    if (in == dfWaveform && gpu_frontend_ && out != dfParams) {
        // -F: only the size is needed to plan the launches; the GPU worker reads the file straight into
        // its context's pinned byte buffer
        struct stat st;
        if (stat(job.src.c_str(), &st) != 0 || !S_ISREG(st.st_mode)) {
            snprintf(msg, sizeof msg, "Can not open waveform file: %s\n", job.src.c_str());
            job.ok = false; job.err = msg;
            return;
        }
        job.file_bytes = (long long)st.st_size;
        job.frames = gpus_[0]->FrontendFrames(job.file_bytes);
        job.cols = nbanks_;
        return;
    }
    if (in == dfWaveform) {
        std::vector<unsigned char> bytes;
        if (!ReadFile(job.src, bytes)) {
            snprintf(msg, sizeof msg, "Can not open waveform file: %s\n", job.src.c_str());
            job.ok = false; job.err = msg;
            return;
        }
        std::vector<float> samples;
        int n = 0;
        DecodeWaveform(bytes, wave_, samples, &n);
        MelBanks mb = mb_proto_;                       // private FFT scratch per call
        mb.Compute(samples, n, job.mel);
        job.frames = mb.NumFrames(n);
        job.cols = nbanks_;
        const float shift = C.GetFloat("framenorm", "shift"), floor_ = C.GetFloat("framenorm", "min_floor");
        if (shift != 0.0f) for (float &v : job.mel) v += shift;                 // srec.cpp:1594-1620
        if (floor_ != -9999.9f) for (float &v : job.mel) if (v < floor_) v = floor_;
    } else {
        std::vector<float> data;
        int rows = 0, cols = 0;
        if (!LoadHTK(job.src, data, &rows, &cols)) {
            snprintf(msg, sizeof msg, "Can not open file: %s\n", job.src.c_str());
            job.ok = false; job.err = msg;
            return;
        }
        job.frames = rows;
        if (in == dfParams) {
            if (cols < nbanks_) { job.ok = false; job.err = "Invalid dimensionality of parameter vectors\n"; return; }
            job.mel.resize((size_t)rows * nbanks_);
            for (int r = 0; r < rows; r++) memcpy(&job.mel[(size_t)r * nbanks_], &data[(size_t)r * cols], sizeof(float) * nbanks_);
            job.cols = nbanks_;
        } else {
            job.post.swap(data);
            job.cols = cols;
        }
    }
    // sentence normalisation happens after the `-t par` exit (srec.cpp:973-974,999)
    if ((in == dfWaveform || in == dfParams) && out != dfParams && sent_mean_norm_ && job.frames > 0)
        SentenceMeanNorm(job.mel.data(), job.frames, nbanks_);
}

static lcrc_softening DeviceSoftening(const std::string &f, const float *a)
{
    lcrc_softening s = {LCRC_SOFT_NONE, a[0], a[1], a[2]};
    if (f == "log") s.func = LCRC_SOFT_LOG;
    else if (f == "igor") s.func = LCRC_SOFT_IGOR;
    else if (f == "gmm_bypass") s.func = LCRC_SOFT_GMM_BYPASS;
    return s;
}

void SpeechRec::Stage3(DataFormat out, Job &job, bool mlf, float *post, int cols, bool device_done)
{
    char msg[1200];
    if (out == dfParams) {
        if (!SaveHTK(job.tgt, job.mel.data(), job.frames, nbanks_)) {
            snprintf(msg, sizeof msg, "Can not create file: %s\n", job.tgt.c_str());
            job.ok = false; job.err = msg;
        }
        return;
    }
    if (out == dfPosteriors) {
        if (!(device_done ? SaveHTKRaw(job.tgt, post, job.frames, cols) : SaveHTK(job.tgt, post, job.frames, cols))) {
            snprintf(msg, sizeof msg, "Can not create file: %s\n", job.tgt.c_str());
            job.ok = false; job.err = msg;
        }
        return;
    }
    // strings: decoder softening (log), Viterbi, labels
    const size_t nvals = (size_t)job.frames * cols;
    if (device_done) { }
    else if (dec_soft_ == "log") for (size_t i = 0; i < nvals; i++) post[i] = logf(post[i]);
    else for (size_t i = 0; i < nvals; i++) post[i] = Soften(dec_soft_, post[i], dec_soft_arg_);
    PhnDec dec;
    dec.LoadPhnList(phoneme_list_);
    dec.SetStatesPerPhn(states_per_phn_);
    dec.SetTimePruning(time_pruning_);
    dec.SetWPenalty(wpenalty_);
    dec.Init();
    if (cols < dec.NumPhonemes() * states_per_phn_) {
        job.ok = false;
        job.err = "posterior vectors are shorter than the phoneme list needs\n";
        return;
    }
    for (int r = 0; r < job.frames; r++) dec.ProcessFrame(post + (size_t)r * cols);
    dec.Done();
    EmitLabels(job, mlf, dec.Labels());
}

// label file (phndec.cpp:230,292) or MLF entry (srec.cpp:137-161,1156,1180)
void SpeechRec::EmitLabels(Job &job, bool mlf, const std::vector<Label> &labels)
{
    std::string text;
    if (mlf) {
        text = "\"" + job.tgt + "\"\n";
        for (const Label &l : labels) text += FormatMlfLine(l);
        text += ".\n";
        job.labels.swap(text);
    } else {
        for (const Label &l : labels) text += FormatLabelLine(l);
        FILE *f = fopen(job.tgt.c_str(), "w");
        if (!f) { job.ok = false; job.err = "Can not create file: " + job.tgt + "\n"; return; }
        fputs(text.c_str(), f);
        fclose(f);
    }
}

// ---- lists -----------------------------------------------------------------------------

std::string SpeechRec::LabelNameForMlf(const std::string &file) const
{
    std::string s = file;
    for (char &c : s) if (c == '\\') c = '/';
    s = ChangeFileSuffix(s, C.GetString("labels", "suffix"));
    if (C.GetBool("labels", "remove_path")) s = ChangeFilePath(s, "*");
    return s;
}

bool SpeechRec::ParseLine(const std::string &line, DataFormat out, bool mlf, Job &job)
{
    char f1[1024], f2[1024], sep[256];
    if (sscanf(line.c_str(), "%1023[^ \n\r\t]%255[ \t]%1023[^ \n\r\t]", f1, sep, f2) == 3) {
        job.src = f1; job.tgt = f2;
        return true;
    }
    if (sscanf(line.c_str(), "%1023s", f1) != 1) return Fail("Invalid line in file list: " + line + "\n");
    job.src = f1;
    switch (out) {
    case dfParams: job.tgt = ChangeFileSuffix(f1, C.GetString("params", "suffix")); break;
    // the reference looks up ("traps","suffix"), which its schema lacks, and aborts
    // (srec.cpp:1224); the documented variable is posteriors/suffix
    case dfPosteriors: job.tgt = ChangeFileSuffix(f1, C.GetString("posteriors", "suffix")); break;
    case dfStrings: job.tgt = mlf ? LabelNameForMlf(f1) : ChangeFileSuffix(f1, C.GetString("labels", "suffix")); break;
    default: break;
    }
    return true;
}

// File lists run as a two-stage pipeline over chunks of kChunkFiles files: while the GPU contexts and
// the decoder / writers work on chunk c, the pool already reads (and, without -F, front-ends) the
// files of chunk c + 1.  Outputs appear in list order; a failing file stops the run like the
// reference's sequential loop does (files before it have been written).
bool SpeechRec::RunJobs(DataFormat in, DataFormat out, std::vector<Job> &jobs, FILE *mlf)
{
    const auto t0 = std::chrono::steady_clock::now();
    if (!pool_) {
        const int threads = host_threads_ > 0 ? host_threads_ : UsableCpus();
        pool_.reset(new ThreadPool(threads > 1 ? threads : 0));
    }
    const bool need_gpu = (in == dfWaveform || in == dfParams) && (out == dfPosteriors || out == dfStrings);
    if (need_gpu) {
        if (!traps_enabled_) return Fail("The 'traps' module have to be enabled for generating posteriors\n");
        if (!EnsureGpus()) return false;
    }
    const auto t1 = std::chrono::steady_clock::now();
    stats_.init_seconds += std::chrono::duration<double>(t1 - t0).count();
    const int n = (int)jobs.size();
    for (const Job &j : jobs) Log(j.tgt.empty() ? j.src + "\n" : j.src + " -> " + j.tgt + "\n");

    constexpr int kChunkFiles = 512;
    std::atomic<long long> stage1_us(0);
    auto stage1 = [&](int lo, int hi) {
        const auto s0 = std::chrono::steady_clock::now();
        pool_->ParallelFor(hi - lo, [&](int i) { Stage1(in, out, jobs[lo + i]); });
        stage1_us += std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - s0).count();
    };
    auto first_error = [&](int lo, int hi) -> const Job * {
        for (int i = lo; i < hi; i++) if (!jobs[i].ok) return &jobs[i];
        return nullptr;
    };
    auto stage23 = [&](int lo, int hi) -> bool {
        if (!need_gpu) {
            pool_->ParallelFor(hi - lo, [&](int i) {
                Job &j = jobs[lo + i];
                Stage3(out, j, mlf != nullptr, out == dfParams ? nullptr : j.post.data(), j.cols);
            });
            return true;
        }
        // Consecutive utterances are packed into launches of <= batch_frames_ frames.  The GPU
        // contexts pull launches from one queue (no exchange between GPUs: every context holds all
        // weights).  Per launch: features are gathered straight into the context's pinned staging
        // buffer, and the decoder / HTK writer read the posteriors straight out of it.
        std::vector<std::pair<int, int>> batches;      // [first, last) job index
        for (int i = lo; i < hi;) {
            int j = i, frames = 0;
            while (j < hi && (j == i || frames + jobs[j].frames <= batch_frames_)) frames += jobs[j++].frames;
            batches.emplace_back(i, j);
            i = j;
        }
        std::atomic<int> next(0);
        std::atomic<bool> failed(false);
        std::vector<std::string> errs(gpus_.size());
        std::vector<double> kms(gpus_.size(), 0.0);
        // posterior writer path: both softening functions and the dump's byte order run in the
        // posterior kernel's epilogue; the host only decodes or writes
        {
            lcrc_softening st[2] = {DeviceSoftening(post_soft_, post_soft_arg_), DeviceSoftening(dec_soft_, dec_soft_arg_)};
            for (auto &g : gpus_)
                if (!g->ConfigureOutput(st, out == dfStrings ? 2 : 1, out == dfPosteriors)) return Fail(g->LastError() + "\n");
        }
        // -D: the decoder runs behind the posterior kernel and only labels cross PCIe
        const bool dev_dec = gpu_decoder_ && out == dfStrings;
        std::vector<std::string> phn_names;
        if (dev_dec) {
            PhnDec names;
            if (!names.LoadPhnList(phoneme_list_)) return Fail("Can not open the phoneme list: " + phoneme_list_ + "\n");
            phn_names = names.Names();
        }
        for (auto &g : gpus_)
            if (!g->ConfigureDecoder(dev_dec ? (int)phn_names.size() : 0, states_per_phn_, time_pruning_, wpenalty_, !dev_dec))
                return Fail(g->LastError() + "\n");
        auto device_labels = [&](Traps &tr, int first, int cnt) -> bool {
            const lcrc_label *lab; const int *lfirst, *lcount; int nu = 0;
            if (!tr.LastLabels(&lab, &lfirst, &lcount, &nu) || nu != cnt) return false;
            pool_->ParallelFor(cnt, [&](int k) {
                Job &j = jobs[first + k];
                std::vector<Label> v((size_t)lcount[k]);
                for (int i = 0; i < lcount[k]; i++) {
                    const lcrc_label &l = lab[lfirst[k] + i];
                    v[i] = Label{l.start, l.end, phn_names[l.phn], l.score};
                }
                EmitLabels(j, mlf != nullptr, v);
            });
            return true;
        };
        auto worker = [&](int g) {
            Traps &tr = *gpus_[g];
            std::vector<int> off;
            for (int b; (b = next.fetch_add(1)) < (int)batches.size() && !failed;) {
                const int first = batches[b].first, cnt = batches[b].second - first;
                off.assign(1, 0);
                for (int k = 0; k < cnt; k++) off.push_back(off.back() + jobs[first + k].frames);
                if (gpu_frontend_ && in == dfWaveform) {
                    // raw bytes in, posteriors out: decode, mel-bank front-end, sentence norm and the
                    // three nets all run on the device
                    // the files go straight into the context's pinned byte buffer (copied in parallel)
                    std::vector<long long> bstart(cnt), blen(cnt);
                    long long pos = 0;
                    for (int k = 0; k < cnt; k++) {
                        bstart[k] = pos;
                        blen[k] = jobs[first + k].file_bytes;
                        pos += blen[k] + (blen[k] & 1);
                    }
                    unsigned char *pinned = nullptr;
                    if (!tr.WaveStageBuffer(pos, &pinned)) { errs[g] = tr.LastError(); failed = true; return; }
                    std::atomic<int> bad(-1);
                    pool_->ParallelFor(cnt, [&](int k) {          // files -> pinned memory, in parallel
                        Job &j = jobs[first + k];
                        FILE *f = fopen(j.src.c_str(), "rb");
                        const bool ok = f && (blen[k] == 0 || fread(pinned + bstart[k], 1, (size_t)blen[k], f) == (size_t)blen[k]);
                        if (f) fclose(f);
                        if (!ok) { int e = -1; bad.compare_exchange_strong(e, k); }
                    });
                    if (bad >= 0) {
                        errs[g] = "Can not open waveform file: " + jobs[first + bad].src; failed = true; return;
                    }
                    std::vector<int> foff(cnt + 1);      // posteriors stay in the context's pinned output buffer
                    if (!tr.WaveStageRun(bstart.data(), blen.data(), cnt, nullptr, foff.data())) {
                        errs[g] = tr.LastError(); failed = true; return;
                    }
                    if (off.back() > 0) kms[g] += tr.LastKernelMs();
                    if (dev_dec) {
                        if (off.back() > 0 && !device_labels(tr, first, cnt)) { errs[g] = "device decoder returned no labels"; failed = true; return; }
                        if (off.back() == 0) pool_->ParallelFor(cnt, [&](int k) { EmitLabels(jobs[first + k], mlf != nullptr, {}); });
                        continue;
                    }
                    float *hp = const_cast<float *>(tr.StagedPosteriors());
                    pool_->ParallelFor(cnt, [&](int k) {
                        Job &j = jobs[first + k];
                        float *pp = hp + (size_t)foff[k] * n_out_;
                        j.cols = n_out_;
                        Stage3(out, j, mlf != nullptr, pp, n_out_, true);
                    });
                    continue;
                }
                float *h_mel = nullptr, *h_post = nullptr;
                if (!tr.StageBuffers(off.back(), &h_mel, &h_post)) { errs[g] = tr.LastError(); failed = true; return; }
                pool_->ParallelFor(cnt, [&](int k) {
                    Job &j = jobs[first + k];
                    memcpy(h_mel + (size_t)off[k] * nbanks_, j.mel.data(), j.mel.size() * sizeof(float));
                    std::vector<float>().swap(j.mel);
                });
                if (!tr.StageRun(off.data(), cnt)) { errs[g] = tr.LastError(); failed = true; return; }
                if (off.back() > 0) kms[g] += tr.LastKernelMs();
                if (dev_dec) {
                    if (off.back() > 0 && !device_labels(tr, first, cnt)) { errs[g] = "device decoder returned no labels"; failed = true; return; }
                    if (off.back() == 0) pool_->ParallelFor(cnt, [&](int k) { EmitLabels(jobs[first + k], mlf != nullptr, {}); });
                    continue;
                }
                pool_->ParallelFor(cnt, [&](int k) {
                    Job &j = jobs[first + k];
                    float *post = h_post + (size_t)off[k] * n_out_;
                    j.cols = n_out_;
                    Stage3(out, j, mlf != nullptr, post, n_out_, true);
                });
            }
        };
        std::vector<std::thread> gt;
        for (size_t g = 0; g < gpus_.size(); g++) gt.emplace_back(worker, (int)g);
        for (auto &t : gt) t.join();
        for (size_t g = 0; g < gpus_.size(); g++) {
            if (!errs[g].empty()) return Fail(errs[g] + "\n");
            stats_.gpu_kernel_ms += kms[g];
        }
        for (int i = lo; i < hi; i++) stats_.frames += jobs[i].frames;
        return true;
    };

    bool ok = true;
    stage1(0, std::min(n, kChunkFiles));
    for (int lo = 0; lo < n && ok; lo += kChunkFiles) {
        const int hi = std::min(n, lo + kChunkFiles), nhi = std::min(n, hi + kChunkFiles);
        // A file that cannot be read stops the run where the reference's sequential loop stops
        // (srec.cpp:1280-1284): everything BEFORE it is still computed and written, then the error is reported.
        const Job *bad = first_error(lo, hi);
        const int stop = bad ? (int)(bad - jobs.data()) : hi;
        std::thread ahead;
        if (!bad && hi < n) ahead = std::thread([&, hi, nhi] { stage1(hi, nhi); });
        if (stop > lo) ok = stage23(lo, stop);
        if (ahead.joinable()) ahead.join();
        if (!ok) break;
        for (int i = lo; i < stop; i++) {
            Job &j = jobs[i];
            if (!j.ok) { ok = Fail(j.err); break; }
            if (mlf) fputs(j.labels.c_str(), mlf);
            j = Job();                               // results are out: release the buffers
        }
        if (ok && bad) ok = Fail(bad->err);
    }
    stats_.stage1_seconds += stage1_us.load() * 1e-6;
    if (!ok) return false;
    stats_.files += n;
    stats_.seconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - t1).count();
    return true;
}

bool SpeechRec::ProcessFileListLine(DataFormat in, DataFormat out, const std::string &line)
{
    std::vector<Job> jobs(1);
    if (!ParseLine(line, out, false, jobs[0])) return false;
    return RunJobs(in, out, jobs, nullptr);
}

bool SpeechRec::ProcessFileList(DataFormat in, DataFormat out, const std::string &list, const std::string &mlf_path)
{
    FILE *fl = fopen(list.c_str(), "r");
    if (!fl) return Fail("Can not open the file list: " + list + "\n");
    FILE *mlf = nullptr;
    if (!mlf_path.empty()) {
        mlf = fopen(mlf_path.c_str(), "w");
        if (!mlf) { fclose(fl); return Fail("Can not create the MLF: " + mlf_path + "\n"); }
        fprintf(mlf, "#!MLF!#\n");
    }
    const size_t kChunk = 1024;                       // utterances in flight (bounds host memory)
    char buf[1024];
    bool ok = true, eof = false;
    std::string parse_err;
    while (ok && !eof && parse_err.empty()) {
        std::vector<Job> jobs;
        while (jobs.size() < kChunk) {
            if (!fgets(buf, 1023, fl)) { eof = true; break; }
            Job j;
            // an invalid line stops the list there; the lines before it are processed first (srec.cpp:1246-1290
            // works line by line)
            if (!ParseLine(buf, out, mlf != nullptr, j)) { parse_err = LastError(); break; }
            jobs.push_back(std::move(j));
        }
        if (!jobs.empty()) ok = RunJobs(in, out, jobs, mlf);
    }
    if (ok && !parse_err.empty()) ok = Fail(parse_err);
    if (mlf) fclose(mlf);
    fclose(fl);
    return ok;
}

}  // namespace phnrec
