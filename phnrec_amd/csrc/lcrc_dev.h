// lcrc_dev.h -- device-side parameter blocks shared by the packer, the launcher
// and the kernels of the fused LCRC posterior kernel.
#ifndef PHNREC_LCRC_DEV_H
#define PHNREC_LCRC_DEV_H

#include <hip/hip_runtime.h>

#include "../../include/lcrc_experimental.h"

namespace phnrec {

// A workgroup owns 16*FT consecutive frames (FT 16-frame MFMA column tiles): FT = 2 for launches that
// fill the GPU with 32-frame workgroups, FT = 1 for smaller ones (twice the workgroups).
constexpr int kTrapLen = 31;   // posteriors/length
constexpr int kHalf = 16;      // taps per half context
constexpr int kShift = 15;     // Traps::GetTrapShift
constexpr int kNCoef = 11;     // C0 + 10 DCT coefficients
constexpr int kMaxTrapLen = 255;   // longest posteriors/length the general kernels take

// One MLP in MFMA fragment order (see lcrc_pack.cpp for the element maps).
struct NetDev {
    const float4 *w1p;   // [nht][nkq][64]  A fragments of layer 1 (4 k-steps per float4)
    const float4 *w2p;   // [nht][not][64]  A fragments of layer 2 (4 hidden units per float4)
    const float *b1;     // [nht*16]  zero padded
    const float *b2;     // [not*16]  zero padded
    const float *mean;   // [nkq*16]  pad 0
    const float *dev;    // [nkq*16]  pad 1
    int n_inp, n_hid, n_out;
    int ksteps;          // ceil(n_inp / 4)   MFMA k-steps of layer 1
    int nkq;             // ceil(ksteps / 4)  float4 groups of k-steps
    int nht;             // ceil(n_hid / 16)  hidden tiles
    int n_ot;            // ceil(n_out / 16)  output tiles
    // split-f16 arithmetic (mlp_dev.h HalfLoop; NULL when the model has no such form): fragment pairs (high, low)
    const float4 *w1h;   // [npairs][ns][2 tiles][2 pieces][64]  A fragments of layer 1 (8 f16 k-values per lane)
    const float4 *w2h;   // [npairs][not][2 pieces][64]          A fragments of layer 2 (8 hidden units of the pair)
    int npairs;          // ceil(nht / 2); b1 is padded to 32 * npairs, mean / dev to 32 * ns
    // ... whose operands are scaled by powers of two (mlp_dev.h "Operand scaling"): the biases in the accumulators'
    // scale, and the factors that undo it
    const float *b1h;    // b1 * 2^(e1 + 6)
    const float *b2h;    // b2 * 2^(e2 + 14)
    float h2_sig_descale;   // 2^-(e1 + 6): folded into FEXP's f64 constant
    float h2_out_descale;   // 2^-(e2 + 14): applied to a wave's partial output tile
};

struct LcrcParams {
    NetDev net[3];       // band0 (left context), band1 (right context), merger
    const float *mel;    // [n_rows][nbanks]
    const int *off;      // [n_utts + 1] utterance row offsets
    float *post;         // [n_rows][n_out]
    const float *win;    // [2][16] half-context windows
    const float *costab; // [10][16] cosf(v_k * (j + 0.5f)) exactly as sDCT evaluates it
    float normc;         // sqrtf(2/16)
    int n_utts, n_rows, nbanks;
    int n_ot_slab;       // max n_ot over the three nets (set by lcrc_launch)
    // Row range: only the rows [row_first, row_end) of the n_rows rows are computed (the others are context:
    // streaming strips, chunks of a long file with halos); `post` points at the output of row_first.
    // lcrc_launch fills in row_end = n_rows when it is 0.
    int row_first, row_end;
    // Split-hidden path (small launches; lcrc_launch decides): a frame tile's hidden dimension is spread
    // over `split_*` workgroups of `tps_*` hidden tiles each, partial output tiles meet in `part` and the
    // last arriver of a tile (ticket in `cnt`) finishes it.  Scratch owned by the context.
    int split_b, tps_b, split_m, tps_m;
    int split_hint;      // 0 = automatic, 1 = never split, k = at most k workgroups per tile (lcrc_set_hidden_split)
    float4 *part;        // [workgroup][net (band phase: 2)][n_ot_slab][64] partial output tiles
    float4 *gimg;        // [tile][nkq_merger][64] merger operand images handed from the band to the merger phase
    unsigned *cnt;       // [tile][2] arrival tickets (band phase, merger phase); zero between launches
    int split_cap_wgs;   // capacity of `part` in workgroups (`gimg`, `cnt`: as many tiles)
    int tile_frames;     // 0 = choose by launch size, 16 or 32 = forced (lcrc_set_tile_frames)
    int arith;           // 0 = f32 MFMA, 1 = split-f16 arithmetic (lcrc_set_arithmetic; fused kernel only)
    // posterior writer path (lcrc_output_configure): softening stages and byte order of `post`
    int out_func[2];     // LCRC_SOFT_* per stage (0 = none)
    float out_c[2][4];   // igor: {middle, 1/middle, 1/(1-middle), -} ; out_l: {ln left base, ln right base}
    float out_l[2][2];
    int out_be;          // store big-endian words
    // optional stage outputs (NULL in production)
    float *dbg_in0, *dbg_in1, *dbg_p0, *dbg_p1, *dbg_g;
    // [grid][8 waves][16] s_memtime stamps; only written by the diagnostic build (-DLCRC_STAMPS)
    unsigned long long *stamps;
};

// LDS carve-up (bytes), computed identically on host and device.
struct LdsPlan {
    unsigned mel, rowinfo, tabs, xf, slab23, norms, gf, slab, total;
};

__host__ __device__ inline unsigned lcrc_round16(unsigned v) { return (v + 15u) & ~15u; }

__host__ __device__ inline LdsPlan lcrc_lds_plan(int ft, int nbanks, int nkq_band, int nkq_merger, int n_ot)
{
    LdsPlan p;
    unsigned o = 0;
    const unsigned uft = (unsigned)ft, bm = 16u * uft;
    // Everything up to the end of xf is dead once the band nets' hidden loops are over; the band
    // classifiers' second pair of slabs (they run side by side, see run_net) lies over it.
    p.slab23 = 0;
    p.mel = o;      o += lcrc_round16((bm + 2u * kShift) * nbanks * 4u);
    p.rowinfo = o;  o += 2u * bm * 4u;
    p.tabs = o;     o += (10u * 16u + 2u * 16u) * 4u;
    p.xf = o;       o += 2u * uft * nkq_band * 1024u;    // [net][f][kq][64] float4
    const unsigned two_slabs = 2u * (uft * n_ot * 1024u);
    if (o < two_slabs) o = two_slabs;
    p.norms = o;    o += (4u * 16u * nkq_band + 2u * 16u * nkq_merger) * 4u;   // mean|dev of the 3 nets
    p.gf = o;       o += uft * nkq_merger * 1024u;       // [f][kq][64] float4
    p.slab = o;     o += two_slabs;                      // two slabs of [ot][f][64] float4
    p.total = o;
    return p;
}

// LDS of the split path's merger phase: operand image, four wave slabs, ticket word
__host__ __device__ inline unsigned lcrc_split_merger_lds(int nkq_merger, int n_ot)
{
    return (unsigned)nkq_merger * 1024u + 4u * (unsigned)n_ot * 1024u + 16u;
}

inline int lcrc_n_ot_slab(const NetDev *nets)
{
    int m = nets[0].n_ot;
    if (nets[1].n_ot > m) m = nets[1].n_ot;
    if (nets[2].n_ot > m) m = nets[2].n_ot;
    return m;
}

// ---- the other `posteriors/system` variants (traps_kernels.hip; "next" row f4) ----------------------
// 1BT_DCT: features kernel (C0 / DCT of every band's 31-point trajectory) -> one MLP.
// 1BT / 3BT: features kernel (the trajectories themselves) -> one small MLP per band, whose epilogue
// writes -ln(p) into the merger's input matrix -> merger MLP.  (traps.cpp:220-283,347-358,409-433)
struct TrapsFeatParams {
    const float *mel;    // [n_rows][nbanks]
    const int *off;      // [n_utts + 1] or NULL (one utterance)
    int n_utts, n_rows, nbanks, trap_bands;
    int mode;            // 0 = trajectories [trap_bands][n_rows][L], 1 = C0/DCT rows [n_rows][trap_bands*shift],
                         // 2 = LCRC at any geometry: [2 half contexts][n_rows][nbanks*shift] (general kernel only)
    int use_hamming, add_c0, shift;
    const float *hamming;   // [L]  sWindow_Hamming over ones (dspc.h:162-167)
    const float *costab;    // [shift][n]  cosf(v_k * (j + 0.5f)) as sDCT evaluates it (dspc.h:206-221); n = L, mode 2: half
    float normc;            // sqrtf(2/n)
    float *out;
    // posteriors/length L (Traps::SetTrapLen): 31 runs the kernels written for it; any other length (and mode 2) the
    // general features kernel.  back = L - 1 - (L - 1) / 2: the output frame's tap (taps 0 .. L-1 = frames r - back .. ).
    int trap_len, back, half;   // half = (L - 1) / 2 + 1 (traps.cpp:288)
    const float *win;           // mode 2: [2][half] the half contexts' windows (windows/band{0,1}.window)
};

struct MlpParams {
    NetDev net;
    const float *in;     // row r at in + r*in_ld (n_inp values)
    float *out;          // row r at out + r*out_ld (n_out values)
    long in_ld, out_ld;
    int n_rows;
    // Many nets of one size class in ONE launch (grid.y = net; the band classifiers of 1BT / 3BT): net y is
    // nets_dev[y], reads in + y*in_net_stride and writes out + out_col[y]; lds_nkq / lds_n_ot = maxima over y
    const NetDev *nets_dev;     // NULL: the single net above
    const int *out_col;
    long in_net_stride;
    int n_nets, lds_nkq, lds_n_ot;
    int neg_log;         // 1: store -(x > 0 ? ln x : 0)   (sLn + sMultiplication(-1), traps.cpp:424-425); 2: +ln (LCRC, :458)
    int out_func[2];     // else: softening stages / byte order of lcrc_output_configure
    float out_c[2][4];
    float out_l[2][2];
    int out_be;
    unsigned long long *stamps;   // unused (keeps run_net's diagnostic hooks compiling)
    // 1BT_DCT fused: when dct.mel != NULL the net's input rows are not read from `in` but computed in the
    // kernel from the mel tile -- the C0 / DCT projection of every band's clamped 31-point trajectory, exactly
    // the arithmetic of traps_features_kernel mode 1 (traps.cpp:180-283, dspc.h:206-233)
    TrapsFeatParams dct;
    int tile_frames;     // 0 = by launch size, 16 / 32 forced
};

constexpr int kMlpKS = 256, kMlpNOT = 13;   // <= 1024 inputs, <= 208 outputs per net
hipError_t traps_features_launch(const TrapsFeatParams &p, hipStream_t stream);
hipError_t mlp_launch(const MlpParams &p, hipStream_t stream, const char **variant = nullptr);
// 1BT / 3BT in one launch (traps_1bt_kernel); hipErrorNotSupported when no size class holds the model
hipError_t traps_1bt_launch(const MlpParams &p, hipStream_t stream, const char **variant = nullptr);
bool mlp_supports(const NetDev &net);

// ---- phoneme-loop Viterbi decoder (phndec_kernels.hip; "next" row f3, optional) ---------------------
struct PhnDecParams {
    const float *logpost;   // [rows][cols] softened (log) posteriors, resident in HBM
    const int *off;         // [n_utts + 1]
    int n_utts, cols;
    int P, S, prune;        // phonemes, states per phoneme, decoder/time_pruning
    float wpen;             // decoder/wpenalty
    lcrc_label *labels;     // [rows]: the labels of utterance u start at labels[off[u]]
    int *count;             // [n_utts]
};
hipError_t phndec_launch(const PhnDecParams &p, hipStream_t stream);
hipError_t phndec_preload_code();

// launcher (lcrc_kernels.hip)
hipError_t lcrc_launch(const LcrcParams &p, hipStream_t stream, const char **variant_name);
hipError_t lcrc_preload_code();
// scratch the split-hidden path needs for `wgs` workgroups (bytes of part / gimg / cnt); 0,0,0 when the
// model has no split kernels
void lcrc_split_scratch(const NetDev *nets, int wgs, size_t *part_bytes, size_t *gimg_bytes, size_t *cnt_bytes);
constexpr int kSplitCapWgs = 512;    // workgroups of a split launch (tiles x split) never exceed this
// variant that WOULD be selected for these nets (no launch); NULL if unsupported
const char *lcrc_variant_for(const NetDev *nets, int nbanks, unsigned *lds_bytes);
// whether split-f16 kernels exist for the model's shape (a shipped shape)
bool lcrc_has_split_f16(const NetDev *nets);

}  // namespace phnrec
#endif
