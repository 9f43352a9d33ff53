/*
 * traps_oracle.h -- CPU restatement of the OTHER `posteriors/system` variants of PhnRec's Traps class
 * (1BT_DCT -- the schema default, srec.cpp:69 --, 1BT and 3BT; the LCRC variant is lcrc_oracle.h).
 *
 * TEST INFRASTRUCTURE ONLY (same rules as lcrc_oracle.h: only tests/, smoke() and bench.py's cpu_baseline
 * may load it, as the checker).
 *
 * Parity status: PINNED in-process against the reference's own Traps class (oracle/_ref/libphnrec_ref.so
 * through ref_shim.cpp, tests/test_oracle.py) on seeded synthetic models -- the reference ships no model and
 * holds no golden vector for these systems; the fixtures under tests/golden/systems/ are outputs of the
 * reference build on those synthetic models (tools/make_golden.py).
 */
#ifndef TRAPS_ORACLE_H
#define TRAPS_ORACLE_H

#include "lcrc_oracle.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_traps orc_traps;

/* Traps::SetSystem / SetNBanks / SetAddC0 / SetHamming / Init (traps.cpp:572-586, 88-171), length = 31.
 * system: "1BT_DCT" (merger only), "1BT" (nbanks band nets), "3BT" (nbanks - 2 band nets). */
int  orc_traps_create(orc_traps **out, const char *model_dir, const char *system, int nbanks,
                      int add_c0, int hamming);
/* ... at any posteriors/length (Traps::SetTrapLen), and for system "LCRC" as well (two band nets, windows from files,
 * traps.cpp:285-343,435-461) -- the LCRC restatement with run-time geometry, incl. add_c0 = 0; lcrc_oracle.h is the one
 * for the shipped geometry (length 31) and also knows the streaming form. */
int  orc_traps_create_geometry(orc_traps **out, const char *model_dir, const char *system, int nbanks,
                               int add_c0, int hamming, int trap_len);
void orc_traps_destroy(orc_traps *t);
int  orc_traps_num_outputs(const orc_traps *t);
int  orc_traps_num_band_nets(const orc_traps *t);
/* post[r] = F(mel[clamp(r-15..r+15)]) (length 31; in general r - (L-1 - (L-1)/2) .. + L - 1) per utterance b = rows [off[b], off[b+1]) -- what ProcessOffline's
 * prime / main / flush sequence yields for any system (srec.cpp:1035-1059).
 * merger_in (optional, [n][merger inputs]) receives the merger's input rows. */
void orc_traps_posteriors_batch(const orc_traps *t, const float *mel, const int *off, int n_utts,
                                float *post, float *merger_in);

#ifdef __cplusplus
}
#endif
#endif
