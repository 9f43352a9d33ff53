/*
 * lcrc_experimental.h -- tuning switches and test hooks of libphnrec_lcrc.so.  Nothing here is needed by a binding of
 * Traps (lcrc.h) or by the list pipeline (lcrc_pipeline.h); the tests, the A/B tools and the CLI's self-checks use them.
 * Same library, same symbols; no stability promise beyond "additions only" within an ABI version.
 */
#ifndef PHNREC_LCRC_EXPERIMENTAL_H
#define PHNREC_LCRC_EXPERIMENTAL_H

#include "lcrc_pipeline.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Test/diagnostic variant of lcrc_posteriors that also returns the stage
 * outputs the reference keeps in Traps::band_input / band_output /
 * merger_input (traps.h:27-29).  Any of the probe pointers may be NULL.
 * in0,in1 [n][nbanks*11] (un-normalised projections), p0,p1 [n][nOut],
 * g [n][2*nOut] (log band posteriors, merger input before its normalisation). */
int lcrc_posteriors_probe(lcrc_ctx *ctx, const float *mel, int n, float *post,
                          float *in0, float *in1, float *p0, float *p1, float *g);
/* y[i] = the front-end's ln() of x[i] in the form named (LCRC_LN_*), computed on device `device_id` (host arrays in and
 * out; no context needed: for self-checks and tests) */
int lcrc_device_ln(int device_id, int form, const float *x, float *y, long long n);
/* Order of the column sums of the sentence mean normalisation (srec.cpp:1500-1511, matrix.h:2101-2116).
 * 1 (default): the reference's sequential f32 sums in frame order, bit for bit (a dependent add chain per
 * utterance and bank: ~13 ns per frame of the longest utterance of the call; utterances run side by side).
 * 0 (opt-in, for very long single utterances): a fixed-shape tree per utterance (256-row blocks from the
 * utterance's first row, strided lane sums folded by halves, block sums added in order) -- deterministic,
 * independent of what else is in the call, a few microseconds for any length; the mean differs from the
 * reference's by ~1e-7 relative. */
int lcrc_set_mean_order(lcrc_ctx *ctx, int sequential);
/* Frames per workgroup: 0 = chosen per launch (whole rounds as pairs of 16-frame workgroups per CU where two fit side by
 * side -- every shipped shape --, else 32-frame ones; 16-frame ones for what fills less than half of the GPU), or 16 / 32
 * forced (tuning and test hook; results are bit-identical either way) */
int lcrc_set_tile_frames(lcrc_ctx *ctx, int frames);
/* NOTE on batch invariance: with the default (0) a frame's last bits depend on the size of the launch it is part
 * of, for every caller of this library (as the reference's do on bunch_size through BLAS's sgemv / sgemm kernels);
 * callers that need bit-identical posteriors however frames are batched set 1, as this repository's CLI does.
 * Small launches (streaming bunches, short utterances: fewer 16-frame tiles than half of the CUs) run
 * on the split-hidden kernels: every frame tile's hidden dimension is spread over several workgroups,
 * whose partial output tiles the last arriver adds in a fixed order.  The result of a frame then depends
 * on the number of workgroups per tile (last bits; each setting is deterministic and within the parity
 * tolerance), i.e. on the size of the launch it is part of.  0 = automatic (default), 1 = never split:
 * every launch uses the fused kernel and a frame's posteriors are bit-identical however it is batched
 * (the CLI sets this); k > 1 = at most k workgroups per tile. */
int lcrc_set_hidden_split(lcrc_ctx *ctx, int workgroups_per_tile);
/* Test hook, inert unless the process environment holds LCRC_FAULT_INJECTION=1 (else LCRC_E_UNSUPPORTED): the nth
 * (0 = next) staging-buffer allocation of this process from now on fails as if the device / pinned memory were
 * exhausted; -1 switches the injection off.  The failing call returns LCRC_E_NOMEM, leaves no half-allocated
 * buffer group behind, and the context stays usable. */
int lcrc_debug_fail_alloc(int nth);
/* The same for launches: the nth (0 = next) posterior launch of this process from now on fails with LCRC_E_DEVICE before
 * anything is queued -- a device that refuses work in the middle of a list (tests of the callers' error paths). */
int lcrc_debug_fail_launch(int nth);

#ifdef __cplusplus
}
#endif
#endif
