// traps_lcrc.cpp -- drop-in replacement for the reference's traps.cpp + nn.cpp: class Traps forwarding to
// the C ABI of include/lcrc.h.  This is the binding INTEGRATION.md shows (the C++ equivalent of a cgo / JNI
// stub); tests/integration/Makefile compiles it together with the reference's OWN srec.cpp / phnrec.cpp /
// melbanks.cpp / phndec.cpp ... where they lie, giving the reference's command line with the MI355X library
// behind its Traps seam.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "traps.h"
#include "lcrc.h"

static lcrc_ctx *H(void *p) { return static_cast<lcrc_ctx *>(p); }

Traps::Traps() : ctx(0), nbanks(15), trap_len(31), useHamming(false), add_c0(true), bunchSize(1), system(stlcrc) {}
Traps::~Traps() { lcrc_destroy(H(ctx)); }

bool Traps::SetSystem(char *sys)                          // traps.cpp:572-586
{
    if (strcmp(sys, "3BT") == 0) system = st3bt;
    else if (strcmp(sys, "1BT") == 0) system = st1bt;
    else if (strcmp(sys, "1BT_DCT") == 0) system = st1bt_dct;
    else if (strcmp(sys, "LCRC") == 0) system = stlcrc;
    else return false;
    return true;
}

void Traps::Init(char *dir)
{
    static const char *names[] = {"3BT", "1BT", "1BT_DCT", "LCRC"};
    lcrc_ctx *h = 0;
    const char *dev = getenv("PHNREC_DEVICE");
    if (lcrc_create_system(&h, dir, names[system], nbanks, trap_len, add_c0 ? 1 : 0, useHamming ? 1 : 0,
                           dev ? atoi(dev) : 0) != LCRC_OK) {
        fprintf(stderr, "%s\n", lcrc_last_error(0));          // the text of traps.cpp:143
        exit(1);
    }
    ctx = h;
}

void Traps::Reset() { lcrc_reset(H(ctx)); }

void Traps::CalcFeaturesBunched(float *be, float *fe, int n, bool needed)
{
    if (lcrc_push(H(ctx), be, n, fe, needed ? 1 : 0) != LCRC_OK) {
        fprintf(stderr, "ERROR: %s\n", lcrc_last_error(H(ctx)));
        exit(1);
    }
}

void Traps::CalcFeatures(float *be, float *fe, int n, bool needed) { CalcFeaturesBunched(be, fe, n, needed); }
int Traps::GetNumOuts() { return lcrc_num_outputs(H(ctx)); }
int Traps::GetDelay() { return lcrc_delay(H(ctx)); }
