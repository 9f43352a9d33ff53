// binding_demo.cpp -- drives class Traps (tests/integration/traps_lcrc.h + traps_lcrc.cpp) the way
// SpeechRec::Init and SpeechRec::ProcessOffline do (srec.cpp:605-624, 1035-1059): setters, Init, Reset,
// prime with 15 frames (neededFea = false), the main part, flush with the last frame repeated.
//   binding_demo MODEL_DIR NBANKS BUNCH mel.f32 post.f32      (raw float32 matrices, row-major)
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <vector>

#include "traps_lcrc.h"

int main(int argc, char **argv)
{
    if (argc != 6) { fprintf(stderr, "usage: %s MODEL_DIR NBANKS BUNCH mel.f32 post.f32\n", argv[0]); return 2; }
    const int nb = atoi(argv[2]), bunch = atoi(argv[3]);
    FILE *f = fopen(argv[4], "rb");
    if (!f) { perror(argv[4]); return 1; }
    std::vector<float> mel;
    float buf[4096];
    size_t got;
    while ((got = fread(buf, sizeof(float), 4096, f)) > 0) mel.insert(mel.end(), buf, buf + got);
    fclose(f);
    const int n = (int)(mel.size() / nb);
    Traps TR;
    char lcrc[] = "LCRC";
    TR.SetSystem(lcrc);
    TR.SetTrapLen(31);
    TR.SetHamming(false);
    TR.SetNBanks(nb);
    TR.SetAddC0(true);
    TR.SetBunchSize(bunch);
    TR.Init(argv[1]);
    const int O = TR.GetNumOuts(), shift = TR.GetTrapShift();
    std::vector<float> post((size_t)n * O), last(mel.end() - nb, mel.end()), pad;
    TR.Reset();
    int primed = std::min(shift, n);
    TR.CalcFeaturesBunched(mel.data(), 0, primed, false);
    for (int i = primed; i < shift; i++) TR.CalcFeaturesBunched(last.data(), 0, 1, false);   // short files, srec.cpp:1045-1048
    if (n > shift) TR.CalcFeaturesBunched(mel.data() + (size_t)shift * nb, post.data(), n - shift, true);
    const int tail = std::min(shift, n);
    for (int i = 0; i < tail; i++) pad.insert(pad.end(), last.begin(), last.end());
    TR.CalcFeaturesBunched(pad.data(), post.data() + (size_t)(n - tail) * O, tail, true);
    f = fopen(argv[5], "wb");
    if (!f) { perror(argv[5]); return 1; }
    fwrite(post.data(), sizeof(float), post.size(), f);
    fclose(f);
    printf("frames %d outputs %d delay %d\n", n, O, TR.GetDelay());
    return 0;
}
