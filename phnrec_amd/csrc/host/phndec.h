// phndec.h -- phoneme-loop Viterbi decoder of the shipped configs (decoder/type=phndec).
// Stays on the host (BASELINE.json north_star).  Behaviour follows the reference's PhnDec
// (phndec.cpp): S-state left-to-right models, ln 0.5 self/next transitions, insertion
// penalty at every phoneme entry, labels emitted when a phoneme boundary falls exactly at
// the `time_pruning` horizon, the rest traced back at Done().
#ifndef PHNREC_HOST_PHNDEC_H
#define PHNREC_HOST_PHNDEC_H

#include <cstdint>
#include <string>
#include <vector>

namespace phnrec {

// std::vector-like array of PODs whose data() is 64-byte aligned: the vector forms of the decoder move whole cache
// lines, and a line-crossing 64-byte access costs two
template <typename T>
class AlignedArray {
public:
    void assign(size_t n, T v)
    {
        store_.assign(n + 64 / sizeof(T), v);
        const uintptr_t a = reinterpret_cast<uintptr_t>(store_.data());
        p_ = reinterpret_cast<T *>((a + 63) & ~uintptr_t(63));
        n_ = n;
    }
    T *data() { return p_; }
    const T *data() const { return p_; }
    size_t size() const { return n_; }
    T &operator[](size_t i) { return p_[i]; }
    const T &operator[](size_t i) const { return p_[i]; }

private:
    std::vector<T> store_;
    T *p_ = nullptr;
    size_t n_ = 0;
};

struct Label {
    int start, end;          // frames
    std::string phn;
    float score;
};

class PhnDec {
public:
    bool LoadPhnList(const std::string &path);            // one symbol per line (phndec.cpp:305-349)
    void SetPhonemes(const std::vector<std::string> &names) { phn_ = names; }
    void SetStatesPerPhn(int n) { S_ = n; }
    void SetTimePruning(int n) { prune_ = n; }
    void SetWPenalty(float p) { wpen_ = p; }
    int NumPhonemes() const { return (int)phn_.size(); }
    const std::vector<std::string> &Names() const { return phn_; }
    void Init();                                          // phndec.cpp:44-94
    void ProcessFrame(const float *logpost);              // phndec.cpp:160-167
    void Done();                                          // phndec.cpp:236-303
    const std::vector<Label> &Labels() const { return labels_; }

private:
    void TimePruning();
    void PruneFrom(int bi, int bj);
    void ProcessFrame512(const float *logpost);
    void Unpack();
    // AVX-512 form (phndec.cpp): (entry winner + 1) << 24 | length per token slot, the uniform entry row as scalars
    bool packed_ = false;
    AlignedArray<int> pk_;          // [S+1][Pp]
    float entry_a_ = 0.0f;
    int entry_prev_ = -1;
    std::vector<std::string> phn_;
    int S_ = 1, prune_ = 50, nframes_ = 0;
    float wpen_ = 0.0f, prev_alpha_ = 0.0f;
    // Token slots state-major: slot (state j, phoneme i) at j * Pp_ + i, Pp_ = the phoneme count padded to the vector
    // width -- the state update is then one straight run over the phonemes per state (compare, select, add: it
    // vectorises), with the same f32 additions and the same tie rules as the reference's phoneme-major loops.
    // Pad slots hold -FLT_MAX and never win a strict comparison.
    int Pp_ = 0;
    AlignedArray<float> alpha_;     // [S+1][Pp]
    AlignedArray<int> prev_, len_;  // [S+1][Pp]
    AlignedArray<float> obs_;       // [S][Pp] this frame's log-posteriors, state-major
    // history of the network-level winners of the last prune+1 frames: a ring, slot of column c = (hpos_ + c) % cols
    std::vector<int> hphn_, hlen_;
    std::vector<float> halpha_;
    int hpos_ = 0;
    std::vector<Label> labels_;
};

// Label text exactly as the reference prints it.
// direct label file: "%d00000 %d00000 %s %f\n" (phndec.cpp:230,292) -> "000000 ..." at frame 0
std::string FormatLabelLine(const Label &l);
// MLF entry: "0" for zero, "%u00000" otherwise (SpeechRec::OnWordMLF, srec.cpp:137-161)
std::string FormatMlfLine(const Label &l);

}  // namespace phnrec
#endif
