/*
 * lcrc_oracle.h -- CPU restatement of PhnRec's LCRC posterior path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the shipped
 * product: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg may load this library, and only as the checker.  The product path
 * (phnrec_amd/csrc, include/lcrc.h) never links, loads or calls it.
 *
 * Parity status: PINNED.  The restatement is checked (tests/test_oracle_*.py)
 * against posteriors dumped by the real reference compiled from
 * /root/reference (oracle/_ref, recipe in oracle/Makefile) for the bundled
 * test.raw on the CZ and EN systems (fixtures under tests/golden/), and
 * in-process against the reference's own Traps / NeuralNet classes through
 * oracle/ref_shim.cpp when oracle/_ref/libphnrec_ref.so is present.
 *
 * Every function cites the reference file:line it follows (paths relative to
 * the reference repository root).
 */
#ifndef LCRC_ORACLE_H
#define LCRC_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_TRAP_LEN 31           /* config: posteriors/length=31, srec.cpp:605-624 */
#define ORC_HALF     16           /* (31-1)/2+1, traps.cpp:93 */
#define ORC_SHIFT    15           /* Traps::GetTrapShift, traps.h:67 */

/* Error codes mirror nn.h:35-42 */
enum { ORC_OK = 0, ORC_NOWEIGHTS = 1, ORC_BADWEIGHTS = 2, ORC_NONORMS = 3,
       ORC_BADNORMS = 4, ORC_MEMORY = 5, ORC_CREATEERR = 6, ORC_WRITEERR = 7 };

/* One 2-layer MLP exactly as NeuralNet holds it after Load (nn.h:58-86):
 * strides padded to a multiple of 4 floats (nn.cpp:633-651). */
typedef struct orc_net {
    int nInp, nHid, nOut;
    int nInp16, nHid16, nOut16;
    float *W1;    /* [nHid16][nInp16]  row = hidden unit              */
    float *W2;    /* [nOut16][nHid16]  row = output unit              */
    float *b1;    /* [nHid16] */
    float *b2;    /* [nOut16] */
    float *mean;  /* [nInp16] pad 0 */
    float *dev;   /* [nInp16] multiplier, pad 1 */
} orc_net;

typedef struct orc_lcrc orc_lcrc;

/* ---- scalar helpers ---------------------------------------------------- */
float orc_fexp(float y);                 /* fexp.h:14-21 (lo word = 0)      */
float orc_fexp_sigmoid(float x);         /* fexp.h:33-38                    */
void  orc_fexp_softmax(int n, float *v); /* fexp.h:49-78                    */

/* ---- one MLP ----------------------------------------------------------- */
int  orc_net_load_nbin(orc_net *net, const char *path);               /* nn.cpp:464-531 */
int  orc_net_load_ascii(orc_net *net, const char *weights, const char *norms); /* nn.cpp:116-462 */
int  orc_net_save_nbin(const orc_net *net, const char *path);         /* nn.cpp:533-592 */
/* Load(): try <weights minus suffix>.nbin, else ASCII (+ write the .nbin cache
 * when write_cache != 0), nn.cpp:594-621 */
int  orc_net_load(orc_net *net, const char *weights, const char *norms, int write_cache);
void orc_net_free(orc_net *net);
/* Forward of n rows, naive loop order (bias first, k ascending, f32):
 * nn.cpp:901-950 -> 872-899 -> 771-793.  in [n][nInp], out [n][nOut]. */
void orc_net_forward(const orc_net *net, const float *in, float *out, int n);
/* Same but also returns the hidden activations [n][nHid] (after sigmoid). */
void orc_net_forward_probe(const orc_net *net, const float *in, float *out,
                           float *hidden, int n);

/* ---- the LCRC estimator ------------------------------------------------ */
/* Traps::Init for system=LCRC, length=31, add_c0=true (traps.cpp:88-171). */
int  orc_lcrc_create(orc_lcrc **out, const char *model_dir, int nbanks);
/* Same from three already-loaded nets + two 16-tap windows (takes ownership). */
int  orc_lcrc_create_from(orc_lcrc **out, int nbanks, orc_net band0, orc_net band1,
                          orc_net merger, const float *win0, const float *win1);
void orc_lcrc_destroy(orc_lcrc *c);
int  orc_lcrc_num_outputs(const orc_lcrc *c);   /* Traps::GetNumOuts traps.h:66 */
int  orc_lcrc_num_inputs(const orc_lcrc *c);    /* band net input size = nbanks*11 */
int  orc_lcrc_nbanks(const orc_lcrc *c);
const orc_net *orc_lcrc_net(const orc_lcrc *c, int which); /* 0,1 band; 2 merger */
const float   *orc_lcrc_window(const orc_lcrc *c, int which);

/* Window/DCT projection of ONE 31-frame context (traps.cpp:285-343).
 * ctx is [nbanks][31] band-major (the be_mat layout, traps.cpp:180-219). */
void orc_lcrc_project(const orc_lcrc *c, const float *ctx, float *in0, float *in1);

/* Stateless whole-utterance form: post[r] = F(mel[clamp(r-15..r+15,0,n-1)])
 * (equals ProcessOffline's prime/main/flush, srec.cpp:1035-1059).
 * mel [n][nbanks] AFTER sentence normalisation; post [n][nOut].
 * Optional probes (may be NULL): in0,in1 [n][nbanks*11]; p0,p1 [n][nOut];
 * g [n][2*nOut]. */
void orc_lcrc_posteriors(const orc_lcrc *c, const float *mel, int n, float *post);
void orc_lcrc_posteriors_probe(const orc_lcrc *c, const float *mel, int n, float *post,
                               float *in0, float *in1, float *p0, float *p1, float *g);
/* Batched form: utterance b = rows [off[b], off[b+1]). */
void orc_lcrc_posteriors_batch(const orc_lcrc *c, const float *mel, const int *off,
                               int n_utts, float *post);

/* Streaming form with the reference's ring-buffer semantics
 * (Traps::Reset / CalcFeaturesBunched, traps.cpp:174-219,470-535). */
void orc_lcrc_reset(orc_lcrc *c);
void orc_lcrc_push(orc_lcrc *c, const float *mel, int n, float *post, int needed, int bunch);
int  orc_lcrc_delay(const orc_lcrc *c);
/* ProcessOffline's par->post block driven through the streaming form
 * (srec.cpp:1035-1059), for cross-checking the stateless form. */
void orc_lcrc_process_offline(orc_lcrc *c, const float *mel, int n, float *post, int bunch);

/* Sentence mean normalisation in place (srec.cpp:1500-1511, matrix.h:2101-2116) */
void orc_sentence_mean_norm(float *mel, int n, int nbanks);

/* PhnDec, the phoneme-loop Viterbi decoder (phndec_oracle.c; phndec.cpp:44-303): one utterance of T rows of
 * softened (log) posteriors; returns the number of labels written (capacity T each). */
int orc_phndec(const float *logpost, int T, int cols, int P, int S, int prune, float wpen,
               int *start, int *end, int *phn, float *score);

/* Multi-threaded whole-utterance posteriors (frames split over threads; each
 * frame is independent).  Used only for bench.py's all-cores cpu_baseline. */
void orc_lcrc_posteriors_mt(const orc_lcrc *c, const float *mel, int n, float *post, int threads);

#ifdef __cplusplus
}
#endif
#endif
