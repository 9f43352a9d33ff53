// ref_shim.cpp -- extern "C" handles onto the REAL reference classes.
//
// TEST INFRASTRUCTURE ONLY.  This file contains no reference code: it includes
// the reference's own headers from /root/reference (never copied into this
// repository) and is compiled together with the reference's traps.cpp / nn.cpp
// where they lie, by oracle/Makefile, into oracle/_ref/libphnrec_ref*.so.
// It lets tests/ and bench.py's cpu_baseline leg call Traps::CalcFeaturesBunched
// (traps.h:61-64) and NeuralNet::Forward (nn.h:52-56) in-process, so the C
// restatement (lcrc_oracle.c) and the HIP path are checked against the
// reference's own arithmetic, not just against its dumped files.
#include <cstring>
#include <cstdlib>

#include "traps.h"

namespace {

// Traps keeps its intermediates protected (traps.h:23-33); a derived class may
// read them.  Valid for the frames of the most recent CalcFeatures bunch.
struct TrapsProbe : public Traps
{
    const float *BandInput(int net) const { return band_input[net]; }
    const float *BandOutput(int net) const { return band_output[net]; }
    const float *MergerInput() const { return merger_input; }
    int BandInputSize(int net) { return band_classifier[net].GetInputSize(); }
    int BandOutputSize(int net) { return band_classifier[net].GetOutputSize(); }
    int MergerInputSize() { return merger.GetInputSize(); }
    int Bunch() const { return bunchSize; }
};

}  // namespace

extern "C" {

// SpeechRec::Init's setter sequence for system=LCRC (srec.cpp:605-624), then
// Traps::Init(dir).  Init exit(1)s on a bad model directory (traps.cpp:141-145).
void *refshim_traps_create_geometry(const char *dir, const char *system, int nbanks, int bunch, int add_c0,
                                    int hamming, int trap_len)
{
    TrapsProbe *t = new TrapsProbe;
    char *sys = strdup(system);
    const bool known = t->SetSystem(sys);
    free(sys);
    if (!known) { delete t; return 0; }
    t->SetTrapLen(trap_len);
    t->SetHamming(hamming != 0);
    t->SetNBanks(nbanks);
    t->SetAddC0(add_c0 != 0);
    t->SetBunchSize(bunch);
    char *d = strdup(dir);
    t->Init(d);
    free(d);
    return t;
}

void *refshim_traps_create_system(const char *dir, const char *system, int nbanks, int bunch, int add_c0,
                                  int hamming)
{
    return refshim_traps_create_geometry(dir, system, nbanks, bunch, add_c0, hamming, 31);
}

void *refshim_traps_create(const char *dir, int nbanks, int bunch)
{
    return refshim_traps_create_system(dir, "LCRC", nbanks, bunch, 1, 0);
}

void refshim_traps_destroy(void *h) { delete static_cast<TrapsProbe *>(h); }
void refshim_traps_reset(void *h) { static_cast<TrapsProbe *>(h)->Reset(); }
int refshim_traps_num_outs(void *h) { return static_cast<TrapsProbe *>(h)->GetNumOuts(); }
int refshim_traps_shift(void *h) { return static_cast<TrapsProbe *>(h)->GetTrapShift(); }
int refshim_traps_delay(void *h) { return static_cast<TrapsProbe *>(h)->GetDelay(); }

void refshim_traps_calc_bunched(void *h, const float *mel, float *post, int n, int needed)
{
    static_cast<TrapsProbe *>(h)->CalcFeaturesBunched(const_cast<float *>(mel), post, n, needed != 0);
}

// One call of at most `bunch` frames, then copy out the intermediates.
// which: 0/1 band_input[0/1], 2/3 band_output[0/1], 4 merger_input (after sLn).
int refshim_traps_probe(void *h, int which, float *dst, int nframes)
{
    TrapsProbe *t = static_cast<TrapsProbe *>(h);
    const float *src;
    int w;
    switch (which) {
    case 0: case 1: src = t->BandInput(which); w = t->BandInputSize(which); break;
    case 2: case 3: src = t->BandOutput(which - 2); w = t->BandOutputSize(which - 2); break;
    case 4: src = t->MergerInput(); w = t->MergerInputSize(); break;
    default: return -1;
    }
    if (nframes > t->Bunch()) nframes = t->Bunch();
    memcpy(dst, src, sizeof(float) * (size_t)w * nframes);
    return w;
}

// The whole par->post block of SpeechRec::ProcessOffline (srec.cpp:1035-1059)
// restated over the reference Traps object: prime, main, flush.
void refshim_traps_process_offline(void *h, const float *mel, int n, int nbanks, float *post)
{
    TrapsProbe *t = static_cast<TrapsProbe *>(h);
    const int S = t->GetTrapShift(), O = t->GetNumOuts();
    float *tmp = new float[(size_t)S * nbanks];
    float *scratch = new float[(size_t)S * O];
    t->Reset();
    if (n >= S) {
        t->CalcFeaturesBunched(const_cast<float *>(mel), scratch, S, false);
    } else {
        memcpy(tmp, mel, sizeof(float) * (size_t)n * nbanks);
        for (int i = n; i < S; i++)
            memcpy(tmp + (size_t)i * nbanks, mel + (size_t)(n - 1) * nbanks, sizeof(float) * nbanks);
        t->CalcFeaturesBunched(tmp, scratch, S, false);
    }
    if (n > S)
        t->CalcFeaturesBunched(const_cast<float *>(mel) + (size_t)S * nbanks, post, n - S);
    int m = n > S ? S : n;
    for (int i = 0; i < m; i++)
        memcpy(tmp + (size_t)i * nbanks, mel + (size_t)(n - 1) * nbanks, sizeof(float) * nbanks);
    t->CalcFeaturesBunched(tmp, post + (size_t)(n - m) * O, m);
    delete[] tmp;
    delete[] scratch;
}

// NeuralNet on its own.
void *refshim_nn_load(const char *weights, const char *norms, int bunch, int *rc)
{
    NeuralNet *n = new NeuralNet;
    char *w = strdup(weights), *m = norms ? strdup(norms) : 0;
    int r = n->Load(w, m, bunch);
    free(w);
    free(m);
    if (rc) *rc = r;
    if (r != NN_OK) { delete n; return 0; }
    return n;
}
void refshim_nn_destroy(void *h) { delete static_cast<NeuralNet *>(h); }
void refshim_nn_dims(void *h, int *inp, int *hid, int *out)
{
    NeuralNet *n = static_cast<NeuralNet *>(h);
    *inp = n->GetInputSize(); *hid = n->GetHiddenSize(); *out = n->GetOutputSize();
}
void refshim_nn_forward(void *h, const float *in, float *out, int n)
{
    static_cast<NeuralNet *>(h)->Forward(const_cast<float *>(in), out, n);
}

#ifdef USE_BLAS
int refshim_uses_blas(void) { return 1; }
#else
int refshim_uses_blas(void) { return 0; }
#endif

}  // extern "C"
