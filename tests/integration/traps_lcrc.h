// traps_lcrc.h -- what a maintainer of rampa069/PhnRec puts in place of the reference's traps.h
// (option (a)): class Traps with the public interface SpeechRec uses (the calls of
// srec.cpp:605-624 before Init; Init / Reset / CalcFeaturesBunched / GetNumOuts / GetTrapShift / GetDelay
// afterwards), its scratch buffers and three NeuralNet members replaced by one opaque handle of the MI355X
// library.  Everything is defined out of line in traps_lcrc.cpp.
//
// It carries the reference header's include guard: tests/integration/Makefile forces this file in front of
// the reference's translation units with -include, which turns their own `#include "traps.h"` into a no-op.
#ifndef TRAPS_H
#define TRAPS_H

struct lcrc_ctx;

class Traps
{
public:
    Traps();
    ~Traps();

    // set-up; call before Init
    bool SetSystem(char *sys);          // "LCRC", "1BT_DCT", "1BT", "3BT"
    void SetNBanks(int v);
    void SetTrapLen(int v);
    void SetAddC0(bool v);
    void SetHamming(bool ham);
    void SetBunchSize(int v);           // grouping only: never changes a value
    void Init(char *dir);               // exit(1) with the reference's message when the model cannot be loaded

    // per file
    void Reset();
    void CalcFeaturesBunched(float *band_energies, float *features, int n = 1, bool neededFea = true);
    void CalcFeatures(float *band_energies, float *features, int n = 1, bool neededFea = true);

    int GetNumOuts();
    int GetTrapShift();
    int GetDelay();

private:
    lcrc_ctx *handle_;
    const char *system_name_;
    int banks_, length_, bunch_;
    bool hamming_, c0_;
};

#endif
