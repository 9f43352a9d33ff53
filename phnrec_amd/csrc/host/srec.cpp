// srec.cpp -- see srec.h
#include "srec.h"
#include "veclog.h"

#include <fcntl.h>
#include <sched.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
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
    CPU_ZERO(&affinity_);
    for (int i = 0; i < n; i++) workers_.emplace_back(new Worker);
    for (int i = 0; i < n; i++) threads_.emplace_back([this, i] { Run(i); });
}

ThreadPool::~ThreadPool()
{
    {
        std::lock_guard<std::mutex> l(mu_);
        stop_ = true;
        WakeLocked((int)workers_.size());
    }
    for (auto &t : threads_) t.join();
}

void ThreadPool::SetAffinity(const cpu_set_t &set)
{
    if (CPU_COUNT(&set) == 0) return;
    std::lock_guard<std::mutex> l(mu_);
    affinity_ = set;
    affinity_gen_++;
}

void ThreadPool::WakeLocked(int k)
{
    while (k-- > 0 && !idle_.empty()) {
        Worker &w = *workers_[(size_t)idle_.back()];
        idle_.pop_back();
        w.wake = true;
        w.cv.notify_one();
    }
}

void ThreadPool::FinishLocked(const Task &t)
{
    if (--t.group->pending == 0) t.group->cv.notify_all();
}

void ThreadPool::Run(int id)
{
    Worker &w = *workers_[(size_t)id];
    std::unique_lock<std::mutex> l(mu_);
    for (;;) {
        if (w.affinity_seen != affinity_gen_) {
            w.affinity_seen = affinity_gen_;
            (void)sched_setaffinity(0, sizeof affinity_, &affinity_);
        }
        if (!queue_.empty()) {                            // a blocked caller's chunks come first
            const Task t = queue_.front();
            queue_.pop_front();
            l.unlock();
            for (int i = t.begin; i < t.end; i++) (*t.fn)(i);
            l.lock();
            FinishLocked(t);
            continue;
        }
        if (!background_.empty()) {
            std::function<void()> bg = std::move(background_.front());
            background_.pop_front();
            l.unlock();
            bg();
            l.lock();
            continue;
        }
        if (stop_) return;
        idle_.push_back(id);
        w.wake = false;
        w.cv.wait(l, [&w] { return w.wake; });
    }
}

void ThreadPool::Submit(std::function<void()> fn)
{
    if (threads_.empty()) { fn(); return; }
    std::lock_guard<std::mutex> l(mu_);
    background_.push_back(std::move(fn));
    WakeLocked(1);
}

void ThreadPool::ParallelFor(int n, const std::function<void(int)> &fn, int grain)
{
    if (n <= 0) return;
    const int max_chunks = grain > 1 ? (n + grain - 1) / grain : n;
    // a few chunks per worker keeps the tail short without flooding the queue
    const int chunks = std::max(1, std::min(max_chunks, 4 * ((int)threads_.size() + 1)));
    if (threads_.empty() || chunks == 1) {
        for (int i = 0; i < n; i++) fn(i);
        return;
    }
    Group g;
    g.pending = chunks;
    std::unique_lock<std::mutex> l(mu_);
    for (int c = 0; c < chunks; c++)
        queue_.push_back(Task{&fn, (int)((long long)n * c / chunks), (int)((long long)n * (c + 1) / chunks), &g});
    WakeLocked(chunks - 1);                                // the caller takes a share itself
    while (g.pending > 0) {
        if (!queue_.empty()) {                             // (possibly another caller's chunk: all the same to the pool)
            const Task t = queue_.front();
            queue_.pop_front();
            l.unlock();
            for (int i = t.begin; i < t.end; i++) (*t.fn)(i);
            l.lock();
            FinishLocked(t);
        } else {
            g.cv.wait(l);
        }
    }
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
    sent_max_norm_ = C.GetBool("offlinenorm", "sent_max_norm");
    sent_chmax_norm_ = C.GetBool("offlinenorm", "sent_chmax_norm");
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
        const int length = C.GetInt("posteriors", "length");
        if (length < 2 || length > 255) return Fail("posteriors/length must lie in 2..255\n");
        // host-only validation of the model directory (the GPU is claimed lazily, when a
        // par -> post conversion is actually requested).  LCRC at the shipped geometry (31 frames, C0, 11 inputs per
        // band): all three nets and the windows; any other geometry or system: the merger (the rest when a context is
        // created).
        int dims[9];
        int info = LCRC_E_UNSUPPORTED;
        if (sys == "LCRC" && length == 31 && C.GetBool("posteriors", "add_c0")) {
            info = lcrc_model_info(config_dir_.c_str(), nbanks_, dims, nullptr, 0, nullptr);
            if (info != LCRC_OK && info != LCRC_E_UNSUPPORTED) {
                snprintf(msg, sizeof msg, "%s\n", lcrc_last_error(nullptr));
                return Fail(msg);
            }
        }
        if (info == LCRC_OK) {
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
        phn_names_ = probe.Names();          // read once: every utterance's decoder starts from this copy
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

// PHNREC_DEVICE_MAP="0,0": the physical device of each of the -g N logical GPUs (default: 0..N-1).  Lets a 1-GPU box
// run the -g 2 arrangement (4 contexts, one launch queue); on an 8-GPU node it picks the GPUs.
bool SpeechRec::DeviceMap(std::vector<int> &devices)
{
    const int n = std::max(1, n_gpus_);
    std::vector<int> dmap;
    if (const char *e = getenv("PHNREC_DEVICE_MAP")) {
        for (const char *q = e; *q;) {
            char *end = nullptr;
            const long v = strtol(q, &end, 10);
            if (end == q || v < 0 || (*end && *end != ','))
                return Fail(std::string("PHNREC_DEVICE_MAP must be a comma-separated list of GPU indices: ") + e + "\n");
            dmap.push_back((int)v);
            q = *end == ',' ? end + 1 : end;
        }
        if ((int)dmap.size() < n) return Fail("PHNREC_DEVICE_MAP names fewer devices than -g asks for\n");
    }
    devices.clear();
    for (int g = 0; g < n; g++) devices.push_back(dmap.empty() ? g : dmap[(size_t)g]);
    return true;
}

void SpeechRec::WarmUpGpuAsync(bool frontend, bool decoder)
{
    if (!warmup_.empty()) return;
    std::vector<int> devices;
    if (!DeviceMap(devices)) { err_.clear(); devices.assign(1, 0); }      // (a bad map is reported where the contexts are planned)
    std::sort(devices.begin(), devices.end());
    devices.erase(std::unique(devices.begin(), devices.end()), devices.end());
    // one thread per distinct device: seven more HIP start-ups of 80-100 ms each would otherwise begin only when the
    // list's workers create their contexts (failures surface in lcrc_create)
    const int what = (frontend ? LCRC_PRELOAD_FRONTEND : 0) | (decoder ? LCRC_PRELOAD_DECODER : 0);
    for (int dev : devices)
        warmup_.emplace_back([dev, what] {
            (void)lcrc_device_warmup(dev);
            if (what) (void)lcrc_device_preload(dev, what);
        });
}

SpeechRec::~SpeechRec()
{
    JoinWarmUp();
}

// Whether lcrc_frontend_configure would take this configuration (its own limits, restated: lin16 / A-law, frames of
// 129..512 samples = FFT 256 / 512, nbanks_full within [max(3, nbanks), 64]).  Asked before -E is switched on by itself.
bool SpeechRec::GpuFrontendTakesConfig()
{
    if (wave_.format != WF_LIN16 && wave_.format != WF_ALAW) return false;
    const int vs = C.GetInt("melbanks", "vector_size"), step = C.GetInt("melbanks", "vector_step");
    if (vs < 129 || vs > 512 || step < 1 || C.GetInt("source", "sample_freq") < 1) return false;
    int nbf = C.GetInt("melbanks", "nbanks_full");
    if (nbf == -1) nbf = nbanks_;
    return nbf >= 3 && nbf >= nbanks_ && nbf <= 64;
}

// One-off set-up of a context behind Init / InitClone: arithmetic, GPU front-end.  Returns "" or the error text
// (called from one thread per GPU, so it does not touch err_).
std::string SpeechRec::SetUpContext(Traps &t)
{
    if (split_f16_ && !t.SetArithmetic(LCRC_ARITH_SPLIT_F16)) return t.LastError() + "\n";
    if (FrontendOn() || EnergiesOn()) {
        if (wave_.noise_level != 0.0f) return "source/noise_level needs the host front-end (libc rand()); drop -F / -E\n";
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
        fe.sent_mean_norm = sent_mean_norm_ && !EnergiesOn() ? 1 : 0;      // (-E: every normalisation runs on the host)
        if (!EnergiesOn()) {
            if (C.GetFloat("framenorm", "shift") != 0.0f || C.GetFloat("framenorm", "min_floor") != -9999.9f)
                return "framenorm/* needs the host front-end; drop -F (or use -E)\n";
            if (sent_max_norm_ || sent_chmax_norm_)
                return "offlinenorm/sent_max_norm and sent_chmax_norm need the host front-end; drop -F (or use -E)\n";
        }
        if (!t.ConfigureFrontend(fe)) return t.LastError() + "\n";
        // ln() as THIS host's libm takes it (glibc's logf sequence, in the build -- fused multiply-adds or not -- that
        // LibmLogfForm() found the process's logf to match): -F's features are then the host front-end's bit for bit.
        // Another libc (0): log() in double rounded once, last bit or one ulp.
        t.SetFrontendLn(LibmLogfForm());
    }
    return std::string();
}

// Contexts: `per_gpu` on each of the -g N GPUs.  A list wants three per GPU (kernel, copy-back and decoding of
// successive launches overlap); one file wants one.  Only the PLACES are made here: every context is built by its own
// worker thread while the list is already running (BringUpContext), so the list starts at its first line and not behind
// a set-up phase (srec.cpp:1246-1290), and the contexts of eight GPUs come up side by side.
bool SpeechRec::PlanGpus(int per_gpu)
{
    const int n = std::max(1, n_gpus_);
    // (PHNREC_CTX_PER_GPU overrides the default for experiments)
    if (const char *e = getenv("PHNREC_CTX_PER_GPU")) per_gpu = std::max(1, std::min(8, atoi(e)));
    if (gpu_devices_.empty() && !DeviceMap(gpu_devices_)) { gpu_devices_.clear(); return false; }
    std::lock_guard<std::mutex> l(ctx_mu_);
    if (gpus_.size() < (size_t)per_gpu * n) {
        gpus_.resize((size_t)per_gpu * n);          // (context g + k * n serves GPU g)
        ctx_state_.resize(gpus_.size(), 0);
    }
    for (size_t i = 0; i < gpus_.size(); i++) ctx_state_[i] = gpus_[i] ? 1 : 0;      // (a failed place of an earlier run is tried again)
    return true;
}

int SpeechRec::FrontendFramesOf(long long file_bytes) const
{
    const long long len = wave_.format == WF_LIN16 ? file_bytes / 2 : file_bytes;
    return len > fe_vector_size_ ? (int)((len - fe_vector_size_) / fe_vector_step_ + 1) : 1;
}

int SpeechRec::BringUpContext(int idx, const ContextPlan &plan, std::string &err, const std::function<void(const char *)> &mark,
                              const std::function<bool()> &worth_it)
{
    const int n = std::max(1, n_gpus_), g = idx % n, k = idx / n;
    // The context whose weights this one shares on the device (lcrc_clone: no file reads, no packing, no upload): the first
    // context of its GPU -- or, for a GPU's first context, that of an earlier logical GPU on the SAME physical device
    // (PHNREC_DEVICE_MAP=0,0,...: one copy of the model per device however many logical GPUs are mapped onto it).
    int base = k > 0 ? g : -1;
    if (k == 0)
        for (int q = 0; q < g; q++)
            if (gpu_devices_[(size_t)q] == gpu_devices_[(size_t)g]) { base = q; break; }
    Traps *t = nullptr;
    std::unique_ptr<Traps> made;
    {
        std::unique_lock<std::mutex> l(ctx_mu_);
        t = gpus_[(size_t)idx].get();
        if (!t && base >= 0) {
            ctx_cv_.wait(l, [&] { return ctx_state_[(size_t)base] != 0; });
            if (ctx_state_[(size_t)base] < 0) {        // failed (its own message is the run's) or left out: so is this one
                const int why = ctx_state_[(size_t)base];
                ctx_state_[(size_t)idx] = why;
                ctx_cv_.notify_all();
                err.clear();
                return why == -2 ? -1 : 0;
            }
        }
    }
    if (!t && base >= 0 && !worth_it()) {
        std::lock_guard<std::mutex> l(ctx_mu_);
        ctx_state_[(size_t)idx] = -2;
        ctx_cv_.notify_all();
        mark("ctx: left out");
        return -1;
    }
    auto failed = [&](const std::string &msg) {
        err = msg;
        std::lock_guard<std::mutex> l(ctx_mu_);
        if (!gpus_[(size_t)idx]) ctx_state_[(size_t)idx] = -1;
        ctx_cv_.notify_all();
        return 0;
    };
    if (!t) {
        made.reset(new Traps);
        t = made.get();
        t->SetSystem(C.GetString("posteriors", "system").c_str());
        t->SetTrapLen(C.GetInt("posteriors", "length"));
        t->SetHamming(C.GetBool("posteriors", "hamming"));
        t->SetNBanks(nbanks_);
        t->SetAddC0(C.GetBool("posteriors", "add_c0"));
        t->SetBunchSize(atoi(C.GetString("posteriors", "bunch_size").c_str()));
        t->SetDevice(gpu_devices_[(size_t)g]);
        // outputs must not depend on how files are packed into launches (-g 1 and -g N write the same bytes)
        t->SetHiddenSplit(1);
        mark(base < 0 ? "ctx: create" : "ctx: clone");
        if (!(base < 0 ? t->Init(config_dir_.c_str()) : t->InitClone(*gpus_[(size_t)base]))) return failed(t->LastError() + "\n");
        mark("ctx: created");
        const std::string e = SetUpContext(*t);
        if (!e.empty()) return failed(e);
        mark("ctx: front-end set");
    }
    // posterior writer path: both softening functions and the dump's byte order run in the posterior kernel's epilogue;
    // -D: the decoder runs behind the posterior kernel and only labels cross PCIe
    if (!t->ConfigureOutput(plan.soft, plan.n_soft, plan.big_endian) ||
        !t->ConfigureDecoder(plan.device_decoder ? (int)phn_names_.size() : 0, states_per_phn_, time_pruning_, wpenalty_, !plan.device_decoder) ||
        !t->SetDecoderOverlap(plan.decoder_overlap))
        return failed(t->LastError() + "\n");
    t->SetLaunchOrder(plan.launch_order);
    t->SetWaitMode(plan.poll_us);
    if (made) {
        // up: clones of this GPU may be made, launches may be taken -- the buffers of full-size launches follow
        std::lock_guard<std::mutex> l(ctx_mu_);
        gpus_[(size_t)idx] = std::move(made);
        ctx_state_[(size_t)idx] = 1;
        ctx_cv_.notify_all();
    }
    mark("ctx: configured");
    if (plan.reserve_rows > 0 && !t->Reserve(plan.reserve_rows, 256, plan.reserve_wave_bytes)) { err = t->LastError() + "\n"; return 0; }
    mark("ctx: reserved");
    return 1;
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

void SpeechRec::Stage1(DataFormat in, DataFormat out, Job &job, bool host_features)
{
    char msg[1200];
    if (in == dfWaveform && (FrontendOn() || EnergiesOn()) && out != dfParams && !host_features) {
        // -F: only the size is needed to plan the launches; the GPU worker reads the file straight into
        // its context's pinned byte buffer
        struct stat st;
        if (stat(job.src.c_str(), &st) != 0 || !S_ISREG(st.st_mode)) {
            snprintf(msg, sizeof msg, "Can not open waveform file: %s\n", job.src.c_str());
            job.ok = false; job.err = msg;
            return;
        }
        job.file_bytes = (long long)st.st_size;
        job.frames = FrontendFramesOf(job.file_bytes);
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
    if ((in == dfWaveform || in == dfParams) && out != dfParams && job.frames > 0) {
        if (sent_mean_norm_) SentenceMeanNorm(job.mel.data(), job.frames, nbanks_);
        if (sent_max_norm_ || sent_chmax_norm_) SentenceMaxNorm(job.mel.data(), job.frames, nbanks_, sent_max_norm_);
    }
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
    dec.SetPhonemes(phn_names_);
    dec.SetStatesPerPhn(states_per_phn_);
    dec.SetTimePruning(time_pruning_);
    dec.SetWPenalty(wpenalty_);
    dec.Init();
    if (cols < dec.NumPhonemes() * states_per_phn_) {
        job.ok = false;
        job.err = "posterior vectors are shorter than the phoneme list needs\n";
        return;
    }
    {
        timespec a, b;
        clock_gettime(CLOCK_THREAD_CPUTIME_ID, &a);
        // The frames are read once, straight out of the context's pinned posterior buffer (DRAM-cold: the copy engine
        // put them there), and a frame's ~250 dependent instructions fill the core's window: the loads of the frames
        // behind it are not issued early enough to hide their latency.  Requested kAhead frames ahead, they are:
        // 10 000-file HU list on an EPYC 9575F, Viterbi CPU seconds 1.0-1.1 -> 0.28-0.30 (33 ns per frame, the rate of
        // the cache-resident micro-benchmark; gpurun_out/r04_exp_prefetch.txt; 4 / 8 / 16 / 32 frames ahead alike).
        constexpr int kAhead = 8;
        const size_t row_bytes = (size_t)cols * sizeof(float);
        const char *base = reinterpret_cast<const char *>(post), *end = base + (size_t)job.frames * row_bytes;
        for (int r = 0; r < job.frames; r++) {
            const char *q = base + (size_t)(r + kAhead) * row_bytes;
            for (const char *e = q + row_bytes; q < e && q < end; q += 64) __builtin_prefetch(q, 0, 3);
            dec.ProcessFrame(post + (size_t)r * cols);
        }
        dec.Done();
        clock_gettime(CLOCK_THREAD_CPUTIME_ID, &b);
        viterbi_ns_ += (long long)(b.tv_sec - a.tv_sec) * 1000000000LL + (b.tv_nsec - a.tv_nsec);
    }
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

}  // namespace phnrec
