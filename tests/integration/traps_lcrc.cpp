// traps_lcrc.cpp -- drop-in replacement for the reference's traps.cpp + nn.cpp: class Traps forwarding to
// the C ABI of include/lcrc.h.  This is the binding INTEGRATION.md shows (the C++ equivalent of a cgo / JNI
// stub); tests/integration/Makefile compiles it together with the reference's OWN srec.cpp / phnrec.cpp /
// melbanks.cpp / phndec.cpp ... where they lie, giving the reference's command line with the MI355X library
// behind its Traps seam.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "traps_lcrc.h"
#include "lcrc.h"

Traps::Traps() : handle_(0), system_name_("LCRC"), banks_(15), length_(31), bunch_(1), hamming_(false), c0_(true) {}
Traps::~Traps() { lcrc_destroy(handle_); }

bool Traps::SetSystem(char *sys)                          // the four names traps.cpp:572-586 accepts
{
    static const char *const known[] = {"3BT", "1BT", "1BT_DCT", "LCRC"};
    for (unsigned i = 0; i < sizeof known / sizeof known[0]; i++)
        if (strcmp(sys, known[i]) == 0) { system_name_ = known[i]; return true; }
    return false;
}
void Traps::SetNBanks(int v) { banks_ = v; }
void Traps::SetTrapLen(int v) { length_ = v; }
void Traps::SetAddC0(bool v) { c0_ = v; }
void Traps::SetHamming(bool ham) { hamming_ = ham; }
void Traps::SetBunchSize(int v) { bunch_ = v; }

void Traps::Init(char *dir)
{
    const char *dev = getenv("PHNREC_DEVICE");
    if (lcrc_create_system(&handle_, dir, system_name_, banks_, length_, c0_ ? 1 : 0, hamming_ ? 1 : 0,
                           dev ? atoi(dev) : 0) != LCRC_OK) {
        fprintf(stderr, "%s\n", lcrc_last_error(0));          // the text of traps.cpp:143
        exit(1);
    }
    // opt-in: the split-f16 arithmetic (include/lcrc.h); a model without that form stays on the f32 kernels
    if (getenv("PHNREC_SPLIT_F16") && lcrc_set_arithmetic(handle_, LCRC_ARITH_SPLIT_F16) != LCRC_OK)
        fprintf(stderr, "WARNING: %s\n", lcrc_last_error(handle_));
}

void Traps::Reset() { lcrc_reset(handle_); }

void Traps::CalcFeaturesBunched(float *band_energies, float *features, int n, bool neededFea)
{
    if (lcrc_push(handle_, band_energies, n, features, neededFea ? 1 : 0) != LCRC_OK) {
        fprintf(stderr, "ERROR: %s\n", lcrc_last_error(handle_));
        exit(1);
    }
}

void Traps::CalcFeatures(float *band_energies, float *features, int n, bool neededFea)
{
    CalcFeaturesBunched(band_energies, features, n, neededFea);
}

int Traps::GetNumOuts() { return lcrc_num_outputs(handle_); }
int Traps::GetTrapShift() { return (length_ - 1) / 2; }
int Traps::GetDelay() { return lcrc_delay(handle_); }
