// phnrec.cpp -- the drop-in command line.  Flags, their meaning, the order of checks and the
// error texts follow the reference's phnrec.cpp:113-299; its private getopt() variant
// (getopt.cpp:22-41: "-xVALUE" or "-x VALUE", bare words skipped) is restated below.
// Additions (letters the reference does not use): -g, -b, -j, -F.
#include <cctype>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>
#include <string>

#include "srec.h"

#include <sys/resource.h>
#include "veclog.h"

using namespace phnrec;

static void Help()
{
    puts("\nUSAGE: phnrec [options]\n");
    puts(" -c dir             configuration directory");
    puts(" -l file            list of files");
    puts(" -i file            input file");
    puts(" -o file            output file");
    puts(" -m file            output MLF");
    puts(" -a                 live audio input");
    puts(" -s fmt [waveform]  source format (wf-waveform, par-parameters, post-posteriors)");
    puts(" -t fmt [strings]   target format (par-parameters, post-posteriors, str-strings)");
    puts(" -w fmt [lin16]     waveform format (lin16, alaw)");
    puts(" -f fmt [str]       live output format (str, strlen, lab)");
    puts(" -p num [-3.8]      phoneme insertion penalty");
    puts(" -v                 verbose");
    puts(" -g num [1]         number of GPUs to spread a file list over (MI355X build)");
    puts(" -b num [32768]     frames per GPU launch when batching a file list (65536 with -D)");
    puts(" -j num [all]       host threads for the front-end and the decoder");
    puts(" -F                 mel-bank front-end on the GPU too (waveform -> posteriors on the device; the host front-end's\n"
         "                    features bit for bit where the host's libm is glibc's: lists take it by themselves)");
    puts(" -E                 the front-end's FFTs and bank sums on the GPU, ln() and the normalisations on the host:\n"
         "                    the host front-end's features bit for bit, at a tenth of its CPU time");
    puts(" -D                 phoneme-loop decoder on the GPU too (only labels leave the device)");
    puts(" -H                 split-f16 arithmetic: f32 products as three exact f16 MFMA products (2x the kernel rate,\n"
         "                    same distance to the reference; shipped LCRC systems)\n");
}

struct Opt {
    int c;
    const char *arg;
};

// One step of the reference's option scanner.  Returns -1 at the end, '?' for an unknown
// option or a missing value, 1 for a bare word.
static int NextOpt(int argc, char **argv, const char *options, int &ind, const char *&arg)
{
    if (++ind == argc) return -1;
    const char *a = argv[ind];
    if (a[0] == '-' && isalpha((unsigned char)a[1])) {
        const char *o = strchr(options, a[1]);
        if (!o) return '?';
        if (o[1] == ':') {
            if (a[2]) arg = a + 2;
            else {
                if (++ind == argc) return '?';
                arg = argv[ind];
            }
        }
        return o[0];
    }
    arg = a;
    return 1;
}

static SpeechRec *g_sr = nullptr;

// peak resident set of this process so far (PHNREC_STATS prints it: pinned buffers count)
static double MaxRssMb()
{
    struct rusage ru;
    return getrusage(RUSAGE_SELF, &ru) == 0 ? ru.ru_maxrss / 1024.0 : 0.0;
}

// the resident set as it stands, by kind (/proc/self/status: RssAnon, RssFile, RssShmem), for PHNREC_STATS
static std::string RssKinds()
{
    std::string out;
    if (FILE *f = fopen("/proc/self/status", "r")) {
        char line[256];
        while (fgets(line, sizeof line, f)) {
            long kb = 0;
            char key[64];
            if (sscanf(line, "%63[^:]: %ld kB", key, &kb) == 2 &&
                (!strcmp(key, "RssAnon") || !strcmp(key, "RssFile") || !strcmp(key, "RssShmem") || !strcmp(key, "VmLck") || !strcmp(key, "VmPin"))) {
                char b[96];
                snprintf(b, sizeof b, " %s_mb=%.1f", key, kb / 1024.0);
                out += b;
            }
        }
        fclose(f);
    }
    return out;
}

static void Die(const std::string &msg)
{
    fprintf(stderr, "ERROR: %s", msg.c_str());
    if (msg.empty() || msg.back() != '\n') fputc('\n', stderr);
    if (g_sr) g_sr->JoinWarmUp();         // never exit() under a thread that is still inside the HIP runtime
    exit(1);
}

int main(int argc, char **argv)
{
    const auto t_main = std::chrono::steady_clock::now();
    const char *config_dir = nullptr, *file_list = nullptr, *input_file = nullptr, *output_file = nullptr;
    const char *output_mlf = nullptr, *wpenalty = nullptr;
    bool live = false, verbose = false, gpu_fe = false, gpu_en = false, gpu_dec = false, split_f16 = false;
    int gpus = 1, batch = 0, threads = 0;
    DataFormat iformat = dfWaveform, oformat = dfStrings;
    WaveFormat wformat = WF_UNKNOWN;

    if (argc == 1) { Help(); return 1; }
    if (argc == 2 && strcmp(argv[1], "--selftest-ln") == 0) {
        // the host front-end's ln() over every non-negative float against this process's libm (veclog.cpp)
        const long long bad = LnSelfTest(0);
        printf("ln(): %s; %lld of 2^31 non-negative values (and a stride of the negative ones) differ from logf()\n", LnForm(), bad);
        return bad == 0 ? 0 : 1;
    }
    if (argc >= 2 && strcmp(argv[1], "--selftest-gpu-ln") == 0) {
        // the GPU front-end's ln() (lcrc_frontend_set_ln, in the form this host's libm matches) over every positive float --
        // and a stride of the other bit patterns -- against this process's logf(): what makes -F's features the host
        // front-end's bit for bit.  phnrec --selftest-gpu-ln [device]
        const int form = LibmLogfForm(), dev = argc > 2 ? atoi(argv[2]) : 0;
        if (form == 0) { printf("this libm's logf matches neither of glibc's two sequences: the GPU front-end uses log() in double\n"); return 2; }
        long long bad = 0, n_all = 0;
        const size_t chunk = 1 << 24;
        std::vector<float> x(chunk), y(chunk);
        for (uint64_t base = 0; base <= 0xffffffffull; base += chunk) {
            const bool positive = base < 0x80000000ull;
            if (!positive && (base >> 24) % 16 != 0) continue;               // the negative half: every sixteenth chunk
            for (size_t k = 0; k < chunk; k++) { const uint32_t u = (uint32_t)(base + k); memcpy(&x[k], &u, 4); }
            if (lcrc_device_ln(dev, form, x.data(), y.data(), (long long)chunk) != LCRC_OK) Die(lcrc_last_error(nullptr));
            for (size_t k = 0; k < chunk; k++) {
                const float want = x[k] > 0.0f ? logf(x[k]) : 0.0f;
                if (memcmp(&want, &y[k], 4) != 0) bad++;
            }
            n_all += (long long)chunk;
        }
        printf("GPU ln(), glibc's logf sequence %s fused multiply-adds: %lld of %lld values (every non-negative float, a sixteenth of "
               "the negative ones) differ from this host's logf()\n", form == 1 ? "with" : "without", bad, n_all);
        return bad == 0 ? 0 : 1;
    }
    int ind = 0;
    for (;;) {
        const char *arg = nullptr;
        const int c = NextOpt(argc, argv, "-c:l:i:o:m:as:t:w:f:p:vg:b:j:FEDH", ind, arg);
        if (c == -1) break;
        switch (c) {
        case 'c': config_dir = arg; break;
        case 'l': file_list = arg; break;
        case 'i': input_file = arg; break;
        case 'o': output_file = arg; break;
        case 'm': output_mlf = arg; break;
        case 'a': live = true; break;
        case 's':
            iformat = ParseDataFormat(arg);
            if (iformat == dfUnknown) Die(std::string("Invalid data format '") + arg + "'. Supported data formats are 'wf', 'mb', 'post' and 'str'.\n");
            break;
        case 't':
            oformat = ParseDataFormat(arg);
            if (oformat == dfUnknown) Die(std::string("Invalid data format '") + arg + "'. Supported data formats are 'wf', 'mb', 'post' and 'str'.\n");
            break;
        case 'w':
            wformat = ParseWaveFormat(arg);
            if (wformat == WF_UNKNOWN) Die(std::string("Invalid waveform format '") + arg + "'. Supported data formats are 'lin16' and 'alaw'.\n");
            break;
        case 'p': wpenalty = arg; break;
        case 'f':
            if (strcmp(arg, "lab") && strcmp(arg, "str") && strcmp(arg, "strlen"))
                Die(std::string("Invalid output format: ") + arg + ". (can be 'lab', 'str', 'strlen')\n");
            break;
        case 'v': verbose = true; break;
        case 'g': gpus = atoi(arg); break;
        case 'b': batch = atoi(arg); break;
        case 'j': threads = atoi(arg); break;
        case 'F': gpu_fe = true; break;
        case 'E': gpu_en = true; break;
        case 'D': gpu_dec = true; break;
        case 'H': split_f16 = true; break;
        case '?': Die("Error during command line parsing\n");
        default: break;                       // bare words are skipped, as in the reference
        }
    }

    // Fault injection for tests of the list pipeline's error paths (inert unless LCRC_FAULT_INJECTION=1: the library refuses
    // to arm its hooks otherwise, and asking without it is an error, not a silent no-op): the n-th staging-buffer
    // allocation / the n-th posterior launch of this process fails (lcrc_debug_fail_alloc, lcrc_debug_fail_launch).
    if (const char *e = getenv("PHNREC_FAIL_ALLOC_NTH"))
        if (lcrc_debug_fail_alloc(atoi(e)) != LCRC_OK) Die(std::string(lcrc_last_error(nullptr)) + "\n");
    if (const char *e = getenv("PHNREC_FAIL_LAUNCH_NTH"))
        if (lcrc_debug_fail_launch(atoi(e)) != LCRC_OK) Die(std::string(lcrc_last_error(nullptr)) + "\n");

    SpeechRec SR;
    g_sr = &SR;
    SR.SetVerbose(verbose);
    if (!config_dir) Die("Configuration directory is not set (-c)\n");
    // a conversion through the posterior estimator will need the GPU: bring the HIP runtime up NOW, on a helper thread,
    // while the configuration, the model files and the weights' re-packing are dealt with on this one
    SR.SetGpus(gpus);
    // (a list takes the GPU front-end by itself, over four or more GPUs the device decoder too; one file keeps the host's unless
    //  the flags say otherwise: the warm-up threads load those kernels' code objects only for runs that will launch them)
    if ((input_file || file_list) && (int)iformat <= (int)dfParams && (int)oformat >= (int)dfPosteriors)
        SR.WarmUpGpuAsync(iformat == dfWaveform && (gpu_fe || gpu_en || file_list != nullptr),
                          oformat == dfStrings && (gpu_dec || (file_list != nullptr && gpus >= 4)));
    if (batch > 0) SR.SetBatchFrames(batch);
    if (threads > 0) SR.SetHostThreads(threads);
    if (gpu_fe && gpu_en) Die("-F and -E are two forms of the GPU front-end: give one\n");
    SR.SetGpuFrontend(gpu_fe);
    SR.SetGpuEnergies(gpu_en);
    SR.SetGpuDecoder(gpu_dec);
    SR.SetSplitF16(split_f16);
    if (!SR.Init(std::string(config_dir) + "/config")) Die(SR.LastError());
    const double config_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_main).count();

    if (wpenalty) {
        float v;
        if (sscanf(wpenalty, "%f", &v) != 1) Die(std::string("Invalid argument for -p switch at command line: ") + wpenalty + "\n");
        SR.SetWPenalty(v);
    }
    if (wformat != WF_UNKNOWN) SR.SetWaveFormat(wformat);
    if (output_file && !input_file) Die("The input file is not specified (-i)\n");
    if (!((int)oformat > (int)iformat)) Die("Unsupported data conversion (-s, -t)\n");

    if (input_file) {
        std::string line = input_file;
        if (output_file) line += std::string(" ") + output_file;
        if (!SR.ProcessFileListLine(iformat, oformat, line)) Die(SR.LastError());
    }
    if (file_list) {
        if (!SR.ProcessFileList(iformat, oformat, file_list, output_mlf ? output_mlf : "")) Die(SR.LastError());
    }
    if (live) Die("live audio input (-a) is outside the scope of the MI355X posterior path\n");

    if (getenv("PHNREC_STATS")) {
        const RunStats &s = SR.Stats();
        // setup_s: what runs in front of the list (pool, plan); wall_s: the list from its first line to its last, the contexts'
        // start-up included -- they come up beside it: first_ctx_s until the first one could take a launch (HIP start-up,
        // model load, pack, upload), create_s until the last one could; first_launch_s: the first launch call of the run
        // (code-object load, cold clock); main_s: since main() was entered
        fprintf(stderr, "phnrec: files=%lld frames=%lld wall_s=%.3f frames_per_s=%.1f xRT=%.6f gpu_kernel_ms=%.3f "
                        "(front_end_s=%.3f setup_s=%.3f first_ctx_s=%.3f create_s=%.3f first_launch_s=%.3f config_s=%.3f main_s=%.3f) "
                        "host_cpu_s=%.3f (stage1=%.3f read=%.3f gather=%.3f decode_write=%.3f viterbi=%.3f) host_threads=%d contexts=%d mode=%s max_rss_mb=%.1f%s\n",
                s.files, s.frames, s.seconds, s.seconds > 0 ? s.frames / s.seconds : 0.0,
                s.frames > 0 ? s.seconds / (s.frames * 0.01) : 0.0, s.gpu_kernel_ms, s.stage1_seconds, s.init_seconds,
                s.first_context_seconds, s.create_seconds, s.first_launch_seconds, config_s,
                std::chrono::duration<double>(std::chrono::steady_clock::now() - t_main).count(),
                s.cpu_stage1 + s.cpu_read + s.cpu_gather + s.cpu_stage3, s.cpu_stage1, s.cpu_read, s.cpu_gather, s.cpu_stage3, s.cpu_viterbi,
                s.host_threads, s.contexts, SR.ModeString().c_str(), MaxRssMb(), RssKinds().c_str());
    }
    // Every output file is closed by now.  Leave without tearing the HIP runtime down piece by piece (contexts, streams,
    // pinned buffers, code objects: tens of milliseconds that a one-file run would notice); the driver reclaims it all.
    // (tools that live on exit handlers -- rocprofv3 and other preloaded profilers -- get the ordinary exit)
    SR.JoinWarmUp();
    fflush(NULL);                            // every open output stream, not only stdout / stderr
#if defined(__SANITIZE_ADDRESS__) || defined(__SANITIZE_THREAD__) || defined(PHNREC_COVERAGE)
    return 0;                                // sanitizer / coverage builds report from exit handlers
#else
    if (getenv("LD_PRELOAD") || getenv("ROCP_TOOL_LIBRARIES") || getenv("HSA_TOOLS_LIB") || getenv("PHNREC_CLEAN_EXIT")) return 0;
    _Exit(0);
#endif
}
