// Host PhnDec (phnrec_amd/csrc/host/phndec.cpp) on random log-posteriors: ns per frame of each form, and a check that
// the three forms (plain / AVX2 / AVX-512) produce identical labels.  g++ -O2 -std=c++17 phndec_host_bench.cpp
// ../../phnrec_amd/csrc/host/phndec.cpp -I../../phnrec_amd/csrc/host -o phndec_host_bench; run with
// PHNREC_NO_AVX512=1 / PHNREC_NO_AVX2=1 to time the other forms.
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "phndec.h"

int main(int argc, char **argv)
{
    const int P = argc > 1 ? atoi(argv[1]) : 61, T = argc > 2 ? atoi(argv[2]) : 900, reps = argc > 3 ? atoi(argv[3]) : 200;
    const int cols = 3 * P + 3;
    std::mt19937 g(5);
    std::vector<float> post((size_t)T * cols);
    // peaky posteriors that move slowly, like a real utterance: a random walk over phonemes
    int cur = 0;
    for (int t = 0; t < T; t++) {
        if (g() % 7 == 0) cur = g() % P;
        float sum = 0;
        for (int c = 0; c < cols; c++) {
            float v = (c / 3 == cur ? 8.0f : 0.0f) + std::uniform_real_distribution<float>(0, 2)(g);
            post[(size_t)t * cols + c] = expf(v);
            sum += post[(size_t)t * cols + c];
        }
        for (int c = 0; c < cols; c++) post[(size_t)t * cols + c] = logf(post[(size_t)t * cols + c] / sum);
    }
    std::vector<std::string> names;
    for (int i = 0; i < P; i++) names.push_back("p" + std::to_string(i));
    size_t nlab = 0;
    double chk = 0;
    const auto t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < reps; r++) {
        phnrec::PhnDec d;
        d.SetPhonemes(names);
        d.SetStatesPerPhn(3);
        d.SetTimePruning(getenv("PRUNE") ? atoi(getenv("PRUNE")) : 40);
        d.SetWPenalty(-2.8125f);
        d.Init();
        for (int t = 0; t < T; t++) d.ProcessFrame(&post[(size_t)t * cols]);
        d.Done();
        nlab = d.Labels().size();
        chk = 0;
        for (auto &l : d.Labels()) chk += l.score * (l.start + 1) + l.end;
    }
    const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    printf("P=%d T=%d: %.1f ns/frame, %zu labels, checksum %.6f\n", P, T, s / reps / T * 1e9, nlab, chk);
    return 0;
}
