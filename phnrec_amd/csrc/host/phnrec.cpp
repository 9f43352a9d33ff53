// phnrec.cpp -- the drop-in command line.  Flags, their meaning, the order of checks and the
// error texts follow the reference's phnrec.cpp:113-299; its private getopt() variant
// (getopt.cpp:22-41: "-xVALUE" or "-x VALUE", bare words skipped) is restated below.
// Additions (letters the reference does not use): -g, -b, -j, -F.
#include <cctype>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "srec.h"

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
    puts(" -b num [32768]     frames per GPU launch when batching a file list (131072 with -D)");
    puts(" -j num [all]       host threads for the front-end and the decoder");
    puts(" -F                 mel-bank front-end on the GPU too (waveform -> posteriors on the device)");
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

static void Die(const std::string &msg)
{
    fprintf(stderr, "ERROR: %s", msg.c_str());
    if (msg.empty() || msg.back() != '\n') fputc('\n', stderr);
    exit(1);
}

int main(int argc, char **argv)
{
    const char *config_dir = nullptr, *file_list = nullptr, *input_file = nullptr, *output_file = nullptr;
    const char *output_mlf = nullptr, *wpenalty = nullptr;
    bool live = false, verbose = false, gpu_fe = false, gpu_dec = false, split_f16 = false;
    int gpus = 1, batch = 0, threads = 0;
    DataFormat iformat = dfWaveform, oformat = dfStrings;
    WaveFormat wformat = WF_UNKNOWN;

    if (argc == 1) { Help(); return 1; }
    int ind = 0;
    for (;;) {
        const char *arg = nullptr;
        const int c = NextOpt(argc, argv, "-c:l:i:o:m:as:t:w:f:p:vg:b:j:FDH", ind, arg);
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
        case 'D': gpu_dec = true; break;
        case 'H': split_f16 = true; break;
        case '?': Die("Error during command line parsing\n");
        default: break;                       // bare words are skipped, as in the reference
        }
    }

    SpeechRec SR;
    SR.SetVerbose(verbose);
    if (!config_dir) Die("Configuration directory is not set (-c)\n");
    SR.SetGpus(gpus);
    if (batch > 0) SR.SetBatchFrames(batch);
    else if (gpu_dec) SR.SetBatchFrames(131072);     // the decoder is sequential per utterance: more of them per launch
    if (threads > 0) SR.SetHostThreads(threads);
    SR.SetGpuFrontend(gpu_fe);
    SR.SetGpuDecoder(gpu_dec);
    SR.SetSplitF16(split_f16);
    if (!SR.Init(std::string(config_dir) + "/config")) Die(SR.LastError());

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
        fprintf(stderr, "phnrec: files=%lld frames=%lld wall_s=%.3f frames_per_s=%.1f xRT=%.6f gpu_kernel_ms=%.3f "
                        "(front_end_s=%.3f setup_s=%.3f)\n",
                s.files, s.frames, s.seconds, s.seconds > 0 ? s.frames / s.seconds : 0.0,
                s.frames > 0 ? s.seconds / (s.frames * 0.01) : 0.0, s.gpu_kernel_ms, s.stage1_seconds, s.init_seconds);
    }
    return 0;
}
