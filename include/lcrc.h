/*
 * lcrc.h -- C ABI of the MI355X LCRC posterior estimator (libphnrec_lcrc.so).
 *
 * This is the drop-in boundary for ONE path of PhnRec: the LCRC
 * split-temporal-context posterior estimator, i.e. everything class Traps
 * (traps.h:21-76) does for posteriors/system=LCRC, including its three
 * NeuralNet forward passes (nn.h:48-56).  The reference has no FFI; the seam is
 * the public interface of Traps as SpeechRec drives it (srec.cpp:605-624 set-up;
 * srec.cpp:1041,1048,1053,1059 offline; srec.cpp:815,856,898 online).  Each
 * entry point below names the reference call it replaces.
 *
 * Conventions: extern "C", plain pointers and sizes, no C++/torch types.
 * Every function returns LCRC_OK (0) or a negative LCRC_E_* code; the message
 * is available from lcrc_last_error().  All frame matrices are row-major
 * float32: mel is [n][nbanks] (log mel-bank energies AFTER sentence
 * normalisation, exactly what SpeechRec hands to Traps), post is
 * [n][lcrc_num_outputs()].  A context is bound to one GPU and one HIP stream;
 * calls on one context must be serialised by the caller (as with Traps, which
 * is not re-entrant); different contexts are independent.
 *
 * There is NO CPU fallback behind this interface: if no gfx950 device is
 * usable, lcrc_create fails with LCRC_E_DEVICE.
 *
 * This header is the CORE: what a binding of Traps needs (INTEGRATION.md routes (a) and (b)).  The entry points of
 * the rows SURVEY 8 marks "next" -- GPU front-end, writer path, decoder on the device -- and what a host that keeps
 * several contexts per GPU in flight uses (this repository's CLI) are in lcrc_pipeline.h; tuning and test hooks in
 * lcrc_experimental.h.  One library, one set of symbols.
 */
#ifndef PHNREC_LCRC_H
#define PHNREC_LCRC_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2: lcrc_clone; sentence mean in the reference's order by default; split-f16 operand images built on request and
 *    scaled (any finite weights); lcrc_debug_fail_alloc needs LCRC_FAULT_INJECTION=1 in the environment */
/* 3: additions only (every version-2 caller links and behaves as before): lcrc_device_pci_bus_id,
 *    lcrc_set_kernel_done_callback, lcrc_wave_stage_energies, lcrc_reserve */
/* 4: additions only: lcrc_set_decoder_overlap, lcrc_prev_labels, lcrc_set_launch_order, lcrc_frontend_set_ln, lcrc_device_ln */
/* 5: additions only: lcrc_debug_fail_launch; the header is split in three (lcrc.h, lcrc_pipeline.h, lcrc_experimental.h):
 *    same symbols, a caller of ABI 4 that included lcrc.h for the moved entry points now includes the header that holds them */
#define LCRC_ABI_VERSION 5

enum {
    LCRC_OK        = 0,
    LCRC_E_ARG     = -1,   /* bad argument (NULL, negative size, wrong geometry)        */
    LCRC_E_IO      = -2,   /* a model file is missing / unreadable (NN_NOWEIGHTS, nn.h) */
    LCRC_E_MODEL   = -3,   /* a model file is malformed or nets are inconsistent       */
    LCRC_E_DEVICE  = -4,   /* no usable HIP device / kernel launch or copy failed      */
    LCRC_E_NOMEM   = -5,
    LCRC_E_UNSUPPORTED = -6 /* a call that exists for the fused LCRC kernels only, nets too large, ... */
};

typedef struct lcrc_ctx lcrc_ctx;

/* ---- life cycle ---------------------------------------------------------- */

/* Replaces the setter block + Traps::Init(dir) (srec.cpp:605-624,
 * traps.cpp:88-171): loads DIR/weights/band{0,1}.{nbin|weights},
 * DIR/norms/band{0,1}.norms, DIR/windows/band{0,1}.window,
 * DIR/weights/merger.*, DIR/norms/merger.norms (path macros config.h:31-39;
 * .nbin preferred, ASCII parsed otherwise, nn.cpp:594-621), re-packs the
 * weights into MFMA fragment order and uploads them to GPU `device_id`.
 * trap_len = 31, add_c0 non-zero and band nets of nbanks * 11 inputs -- the
 * geometry of every shipped system -- run the fused kernels.  Any other geometry
 * Traps accepts (SetTrapLen 2..255, odd or even; SetAddC0(false); another number
 * of coefficients per band, traps.cpp:285-343 as written, incl. its walk over
 * be_mat for even lengths) is computed by general kernels in three launches
 * (features, the two band nets, the merger: lcrc_kernel_name "lcrc_general"):
 * correct to the same tolerance, not tuned; lcrc_posteriors_rows, the stage
 * probes, lcrc_model_info and the split-f16 arithmetic exist for the fused
 * kernels only.  Unlike Traps::Init it returns an error instead of exit(1). */
int lcrc_create(lcrc_ctx **out, const char *model_dir, int nbanks, int trap_len,
                int add_c0, int device_id);
/* A further context for the same model on the same GPU (the reference would construct a second Traps and load the
 * files again, traps.cpp:88-171): own stream, staging buffers, streaming state and settings (all at their defaults),
 * but the read-only device buffers -- packed weights, biases, norms, tables -- are SHARED with `src` and freed with the
 * last context that uses them.  No file is read, nothing is packed or uploaded.  This is how the CLI gets its second
 * context per GPU (one stages / decodes while the other's launch runs). */
int lcrc_clone(lcrc_ctx **out, const lcrc_ctx *src);

/* The other values of posteriors/system ("next" row f4): "1BT_DCT" (the schema default, srec.cpp:69: C0 / DCT
 * of every band's 31-point trajectory into one net), "1BT" (one 31-input net per band, -ln of their outputs
 * into the merger) and "3BT" (as the reference codes it: 1BT over the first nbanks - 2 bands), with
 * posteriors/hamming and posteriors/add_c0 (traps.cpp:88-171,220-283,347-358,409-433), at any posteriors/length in
 * 2..255 (31: one fused launch per batch; other lengths: features / band nets / merger launches); "LCRC" forwards to
 * lcrc_create (which ignores hamming, as the reference does).  All entry points below work on such a
 * context except lcrc_posteriors_probe and lcrc_model_info.  These systems run as separate feature /
 * MLP launches, not as one fused kernel.  lcrc_net_dims: band classifiers first, the merger last. */
int lcrc_create_system(lcrc_ctx **ctx, const char *model_dir, const char *system, int nbanks, int trap_len,
                       int add_c0, int hamming, int device_id);
/* Host-only: Traps::GetNumOuts (the merger's output count, traps.h:66) for any system, or a negative
 * LCRC_E_* code (text in lcrc_last_error(NULL)). */
int lcrc_model_outputs(const char *model_dir, const char *system);
void lcrc_destroy(lcrc_ctx *ctx);

/* Message of the last failing call on `ctx`; with ctx == NULL, of the last
 * failing lcrc_create on this thread.  Never NULL. */
const char *lcrc_last_error(const lcrc_ctx *ctx);

int lcrc_abi_version(void);

/* Host-only pre-flight (no GPU touched): loads and validates the model directory
 * exactly as lcrc_create does and reports the three nets' dims
 * (dims9 = {inp,hid,out} x {band0, band1, merger}), the kernel variant that
 * would run and its LDS footprint.  Any out pointer may be NULL. */
int lcrc_model_info(const char *model_dir, int nbanks, int *dims9, char *kernel,
                    size_t kernel_cap, unsigned *lds_bytes);

/* ---- geometry (Traps getters, traps.h:66-68) ------------------------------- */
int lcrc_num_outputs(const lcrc_ctx *ctx);   /* Traps::GetNumOuts            */
int lcrc_num_banks(const lcrc_ctx *ctx);
int lcrc_trap_shift(const lcrc_ctx *ctx);    /* Traps::GetTrapShift == 15    */
int lcrc_device(const lcrc_ctx *ctx);
/* dims of net `which` (0,1 band classifiers, 2 merger): NeuralNet::Get*Size */
int lcrc_net_dims(const lcrc_ctx *ctx, int which, int *n_inp, int *n_hid, int *n_out);

/* ---- whole-utterance form ---------------------------------------------------
 * Replaces the prime / main / flush sequence of SpeechRec::ProcessOffline
 * (srec.cpp:1035-1059): post[r] = F(mel[clamp(r-15 .. r+15, 0, n-1)]).
 * Host buffers; synchronous (H2D, kernel, D2H).  n == 0 is a no-op. */
int lcrc_posteriors(lcrc_ctx *ctx, const float *mel, int n, float *post);

/* Many utterances in one launch: utterance u occupies rows [off[u], off[u+1])
 * of mel and post (off has n_utts+1 non-decreasing entries, off[0] == 0);
 * contexts never cross an utterance boundary.  Empty utterances are allowed. */
int lcrc_posteriors_batch(lcrc_ctx *ctx, const float *mel, const int *off, int n_utts,
                          float *post);

/* Same, on DEVICE pointers and asynchronous on `hip_stream` (a hipStream_t; NULL
 * = HIP's default stream, as everywhere in HIP).  d_off may be NULL when
 * n_utts == 1 (one utterance of n_rows frames).  Nothing is copied or
 * synchronised; the caller orders the launch against its own work through the
 * stream it passes.  Small launches (see lcrc_set_hidden_split) use scratch
 * buffers of the context: launches of ONE context must not overlap on the
 * device then -- issue them on one stream, or pin the fused kernel. */
int lcrc_posteriors_device(lcrc_ctx *ctx, const float *d_mel, const int *d_off, int n_utts,
                           int n_rows, float *d_post, void *hip_stream);

/* A row range of one utterance (or of a chunk of one, cut with its 15-frame halos): mel holds n_rows
 * frames, only the posteriors of rows [row_first, row_first + row_count) are computed and returned
 * (post has row_count rows); the other rows are context.  This is how lcrc_push evaluates a bunch on
 * [history | pushed frames] and how a file longer than one launch is cut.  posteriors/system=LCRC only. */
int lcrc_posteriors_rows(lcrc_ctx *ctx, const float *mel, int n_rows, int row_first, int row_count,
                         float *post);

/* ---- streaming form (Traps semantics) -----------------------------------------
 * lcrc_reset == Traps::Reset (traps.cpp:174-177).
 * lcrc_push  == Traps::CalcFeaturesBunched(mel, post, n, needed)
 * (traps.cpp:518-535): the first frame after a reset floods the 31-frame
 * history; when `needed`, post row i is the estimate for the window ENDING at
 * pushed frame i (i.e. centred 15 frames earlier); when !needed only the
 * history advances and `post` is not touched (may be NULL).  The frames are kept in a pinned strip
 * the kernel reads in place and only the n pushed rows are computed (no re-upload of the history,
 * no allocation per call).  A push of >= 4096 frames goes through the context's staging buffers instead
 * (explicit copies at PCIe rate) and may grow them: pointers handed out by lcrc_stage_buffers /
 * lcrc_wave_stage_buffer are INVALID after such a push -- ask for them again (a caller that mixes the staged
 * entry points with streaming on ONE context must do so anyway: both use the context's one set of buffers).
 * lcrc_delay == Traps::GetDelay (frames pushed since reset minus one, capped
 * at 9999, traps.cpp:199,215-217). */
int lcrc_reset(lcrc_ctx *ctx);
int lcrc_push(lcrc_ctx *ctx, const float *mel, int n, float *post, int needed);
int lcrc_delay(const lcrc_ctx *ctx);

/* ---- measurement -------------------------------------------------------------
 * Device time of the most recent posterior kernel launch on this context, from
 * HIP events recorded around it on the stream it ran on.  Blocks until that
 * launch has finished. */
int lcrc_last_kernel_ms(lcrc_ctx *ctx, float *ms);
/* Enable/disable the event pair (default on; costs two hipEventRecord per launch) */
int lcrc_set_timing(lcrc_ctx *ctx, int enabled);
/* Arithmetic of the LCRC kernels.  LCRC_ARITH_F32 (default): v_mfma_f32_16x16x4_f32, the reference's f32 products
 * one by one.  LCRC_ARITH_SPLIT_F16: every f32 operand as a (high, low) pair of f16 values and every product as three
 * exact f16 x f16 MFMA products accumulated in f32 (what is dropped is below 2^-22 of a product -- the size of f32's
 * own rounding of the sums; measured distance to the reference: the same few 1e-6 as LCRC_ARITH_F32).  So that the low
 * halves keep their 11 bits (an f16 below 2^-14 is subnormal) every operand is scaled by a power of two before it is
 * split -- each weight matrix so that its largest entry lies in (2^13, 2^14], net inputs by 2^6, hidden activations by
 * 2^14 -- and the accumulators are scaled back exactly; a model with tiny (or huge) weights loses nothing.  The f16
 * MFMA runs 16x the f32 MFMA's rate, so the kernel is 2-3x faster; it exists for the shipped LCRC shapes (else
 * LCRC_E_UNSUPPORTED and the setting is unchanged; also for non-finite weights); normalised net inputs beyond +-1023
 * are clamped there (documented deviation: a feature 1000 standard deviations out saturates every sigmoid anyway).  The
 * operand images are built and uploaded by the first call that asks for them.  Every launch uses the fused kernel in
 * this mode (as with lcrc_set_hidden_split(h, 1)). */
#define LCRC_ARITH_F32 0
#define LCRC_ARITH_SPLIT_F16 1
int lcrc_set_arithmetic(lcrc_ctx *ctx, int arithmetic);
/* NOTE on batch invariance: by default small launches (streaming bunches, short utterances) run on split-hidden kernels and
 * a frame's LAST BITS then depend on the size of the launch it is part of (as the reference's do on bunch_size through
 * BLAS).  lcrc_set_hidden_split(ctx, 1) (lcrc_experimental.h) pins the fused kernel: bit-identical however frames are
 * batched; this repository's CLI sets it. */
/* Name of the kernel variant selected for this model ("cz_42_69_9", "generic_64_104_13", ...; the same in both arithmetics) */
const char *lcrc_kernel_name(const lcrc_ctx *ctx);

#ifdef __cplusplus
}
#endif
#endif
