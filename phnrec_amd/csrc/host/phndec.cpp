// phndec.cpp -- see phndec.h
#include "phndec.h"

#include <cfloat>
#include <cstdio>
#include <cstring>
#include <fstream>

namespace phnrec {

namespace {
const float kLogHalf = -0.69314718055994530941723212145818f;   // both transitions (phndec.cpp:9,14-15)
}

bool PhnDec::LoadPhnList(const std::string &path)
{
    std::ifstream in(path.c_str());
    if (!in) return false;
    phn_.clear();
    std::string line;
    while (std::getline(in, line)) {
        size_t e = line.find_first_of("\r\n");
        if (e != std::string::npos) line.erase(e);
        phn_.push_back(line);
    }
    return true;
}

void PhnDec::Init()
{
    const int P = (int)phn_.size(), W = S_ + 1;
    alpha_.assign((size_t)P * W, -FLT_MAX);
    prev_.assign((size_t)P * W, -1);
    len_.assign((size_t)P * W, 0);
    hphn_.assign(prune_ + 1, -1);
    hlen_.assign(prune_ + 1, -1);
    halpha_.assign(prune_ + 1, -1.0f);
    for (int i = 0; i < P; i++) alpha_[(size_t)i * W] = wpen_;   // entry state carries the penalty
    nframes_ = 0;
    prev_alpha_ = 0.0f;
    labels_.clear();
}

void PhnDec::ProcessFrame(const float *f)
{
    const int P = (int)phn_.size(), W = S_ + 1;
    // inside the models, last state first; pdf of state j of phoneme i is i*S + (j-1)
    for (int i = 0; i < P; i++) {
        float *a = &alpha_[(size_t)i * W];
        int *pv = &prev_[(size_t)i * W], *ln = &len_[(size_t)i * W];
        for (int j = S_; j > 0; j--) {
            const float stay = a[j] + kLogHalf, enter = a[j - 1] + kLogHalf;
            const float obs = f[i * S_ + (j - 1)];
            if (stay > enter) {
                a[j] = stay + obs;
                ln[j] += 1;
            } else {
                a[j] = enter + obs;
                pv[j] = pv[j - 1];
                ln[j] = ln[j - 1] + 1;
            }
        }
    }
    // network level: the best exit token re-enters every phoneme (first strict maximum)
    float best = -FLT_MAX;
    int bi = 0;
    for (int i = 0; i < P; i++) {
        const float t = alpha_[(size_t)i * W + S_];
        if (t > best) { best = t; bi = i; }
    }
    for (size_t k = 1; k < hphn_.size(); k++) {
        hphn_[k - 1] = hphn_[k]; hlen_[k - 1] = hlen_[k]; halpha_[k - 1] = halpha_[k];
    }
    hphn_.back() = prev_[(size_t)bi * W + S_];
    hlen_.back() = len_[(size_t)bi * W + S_];
    halpha_.back() = best;
    for (int i = 0; i < P; i++) {
        alpha_[(size_t)i * W] = best + wpen_;
        prev_[(size_t)i * W] = bi;
        len_[(size_t)i * W] = 0;
    }
    nframes_++;
    TimePruning();
}

// phndec.cpp:191-234
void PhnDec::TimePruning()
{
    const int cols = (int)hlen_.size(), P = (int)phn_.size(), W = S_ + 1;
    if (nframes_ < cols) return;
    float best = -FLT_MAX;
    int blen = 1, bprev = 0;
    for (int i = 0; i < P; i++)
        for (int j = 1; j <= S_; j++)
            if (alpha_[(size_t)i * W + j] > best) {
                best = alpha_[(size_t)i * W + j];
                blen = len_[(size_t)i * W + j];
                bprev = prev_[(size_t)i * W + j];
            }
    int offs = cols - 1 - blen, phn = bprev;
    while (offs > 0) {
        const int l = hlen_[offs];
        phn = hphn_[offs];
        offs -= l;
    }
    if (offs == 0) {                       // a phoneme ends exactly at the pruning horizon
        const int end = nframes_ - cols + 1, start = end - hlen_[0];
        const float like = halpha_[0] - prev_alpha_;
        prev_alpha_ = halpha_[0];
        if (phn >= 0) labels_.push_back(Label{start, end, phn_[phn], like});
    }
}

void PhnDec::Done()
{
    const int cols = (int)hlen_.size();
    int offs = cols - 1, end = nframes_;
    int phn = prev_[0];                    // the winner that entered the loop last
    std::vector<Label> tail;
    while (offs > 0 && phn != -1) {
        const int len = hlen_[offs], start = end - len;
        const float a = halpha_[offs];
        const int pphn = hphn_[offs];
        offs -= len;
        const float like = offs > 0 ? a - halpha_[offs] : a - prev_alpha_;
        tail.push_back(Label{start, end, phn_[phn], like});
        end = start;
        phn = pphn;
    }
    for (size_t i = tail.size(); i-- > 0;) labels_.push_back(tail[i]);
}

std::string FormatLabelLine(const Label &l)
{
    char buf[512];
    snprintf(buf, sizeof buf, "%d00000 %d00000 %s %f\n", l.start, l.end, l.phn.c_str(), l.score);
    return buf;
}

std::string FormatMlfLine(const Label &l)
{
    char a[32], b[32], buf[512];
    if (l.start == 0) snprintf(a, sizeof a, "0"); else snprintf(a, sizeof a, "%u00000", (unsigned)l.start);
    if (l.end == 0) snprintf(b, sizeof b, "0"); else snprintf(b, sizeof b, "%u00000", (unsigned)l.end);
    snprintf(buf, sizeof buf, "%s %s %s %f\n", a, b, l.phn.c_str(), l.score);
    return buf;
}

}  // namespace phnrec
