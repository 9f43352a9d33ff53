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
#define LCRC_ABI_VERSION 4

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
/* Optional: starts the HIP runtime and GPU `device_id`'s context (~0.2 s in a fresh process, by far the largest part of
 * a first lcrc_create) and returns when they are up.  Thread-safe; meant to be called from a helper thread at program
 * start so that the caller's own initialisation -- configuration, the model files and their re-packing inside
 * lcrc_create, the first file's front-end -- overlaps with it (the CLI does).  It also brings the posterior kernels' code
 * object onto the device (20-40 ms that a context's creation or first launch pays otherwise; beside the caller's
 * lcrc_create -- stream, weight upload -- it costs nothing: a one-file `phnrec` run 0.15-0.24 s instead of 0.21-0.29 on
 * the same boxes).  No reference counterpart: the reference has no device to bring up. */
int lcrc_device_warmup(int device_id);
/* PCI address of GPU `device_id` ("0000:c1:00.0") into buf: lets a host that drives several GPUs place the threads of
 * each near it (the CLI pins a GPU's worker threads to the CPUs of /sys/bus/pci/devices/<id>/numa_node).  No reference
 * counterpart. */
int lcrc_device_pci_bus_id(int device_id, char *buf, int len);
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

/* Zero-copy variant of lcrc_posteriors_batch for callers that assemble batches themselves
 * (this repository's SpeechRec does): lcrc_stage_buffers returns pinned host buffers owned
 * by the context with room for `rows` frames (valid until the next lcrc_stage_buffers call
 * with a larger size, or lcrc_destroy); the caller writes mel[rows][nbanks] into *mel,
 * calls lcrc_stage_run (synchronous: the kernel reads the features where they lie and stores
 * the posteriors straight into *post, both over PCIe while it runs -- no copy commands) and
 * reads post[rows][n_out] from *post.  `rows` of lcrc_stage_run is off[n_utts]. */
int lcrc_stage_buffers(lcrc_ctx *ctx, int rows, float **mel, float **post);
int lcrc_stage_run(lcrc_ctx *ctx, const int *off, int n_utts);

/* Test/diagnostic variant of lcrc_posteriors that also returns the stage
 * outputs the reference keeps in Traps::band_input / band_output /
 * merger_input (traps.h:27-29).  Any of the probe pointers may be NULL.
 * in0,in1 [n][nbanks*11] (un-normalised projections), p0,p1 [n][nOut],
 * g [n][2*nOut] (log band posteriors, merger input before its normalisation). */
int lcrc_posteriors_probe(lcrc_ctx *ctx, const float *mel, int n, float *post,
                          float *in0, float *in1, float *p0, float *p1, float *g);

/* ---- waveform entry ("next" row of the path: the mel-bank front-end on the GPU) -----------
 * Replaces, for whole files, the wf -> par block of SpeechRec::ProcessOffline
 * (srec.cpp:939-999): ConvertWaveformFormat, MelBanks::GetFeatures per frame, and the
 * sentence mean normalisation, with the values SpeechRec::Init hands to MelBanks
 * (srec.cpp:537-561).  The FFT, window and mel filters follow the reference operation by
 * operation (ln is evaluated in double and rounded once: last-bit agreement with glibc's logf in
 * all but rare cases).  source/noise_level
 * (libc rand()) has no device equivalent and is not offered. */
typedef struct lcrc_frontend {
    int wave_format;         /* 1 = lin16 (host byte order), 2 = A-law                     source/format      */
    int sample_freq;         /*                                                            source/sample_freq */
    int vector_size;         /* samples per frame (<= 512)                                 melbanks/vector_size */
    int vector_step;         /*                                                            melbanks/vector_step */
    int nbanks_full;         /* -1 = nbanks                                                melbanks/nbanks_full */
    float lower_freq, higher_freq, preem_coef;            /*                               melbanks/...       */
    float scale, dc_shift;   /*                                                            source/scale, dc_shift */
    int z_mean_source;       /*                                                            melbanks/z_mean_source */
    int sent_mean_norm;      /* applied before the posteriors, never to lcrc_wave_to_mel   offlinenorm/sent_mean_norm */
} lcrc_frontend;

int lcrc_frontend_configure(lcrc_ctx *ctx, const lcrc_frontend *cfg);
/* How the front-end takes ln() (sLn, dspc.h:155-160: x > 0 ? logf(x) : 0).  The reference's bits are those of the HOST
 * libm's logf.  glibc's logf (2.28 and later) is a fixed sequence of IEEE double operations, in one of two builds that glibc
 * picks at load time (with fused multiply-adds / without): LCRC_LN_GLIBC_FMA and LCRC_LN_GLIBC run that sequence on the
 * device and give that libm's result for every input -- a caller that has checked which of the two its libm matches (this
 * repository's CLI does: host/veclog.cpp, 300 000 values at start-up; `phnrec --selftest-gpu-ln` compares every positive
 * float) gets features equal to a host front-end's bit for bit.  LCRC_LN_DOUBLE (default): log() in double rounded once --
 * independent of any libm, equal to glibc's result except in the rare cases where glibc's own 0.818-ulp error shows.
 * Holds for every later waveform call on the context (not for lcrc_wave_stage_energies, which stops in front of ln). */
enum { LCRC_LN_DOUBLE = 0, LCRC_LN_GLIBC_FMA = 1, LCRC_LN_GLIBC = 2 };
int lcrc_frontend_set_ln(lcrc_ctx *ctx, int form);
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
/* frames a file of n_bytes yields: len > vs ? (len - vs)/step + 1 : 1   (srec.cpp:945) */
int lcrc_frontend_frames(const lcrc_ctx *ctx, long long n_bytes);
/* bytes: the raw files back to back (no header parsing, like the reference); byte_off[n_utts+1].
 * frame_off (out, [n_utts+1]) receives the first row of each utterance.  mel / post must have
 * room for sum of lcrc_frontend_frames() rows.  lcrc_wave_to_mel returns the features BEFORE
 * sentence normalisation (what `-t par` dumps). */
int lcrc_wave_to_mel(lcrc_ctx *ctx, const unsigned char *bytes, const long long *byte_off, int n_utts,
                     float *mel, int *frame_off);
int lcrc_wave_to_posteriors(lcrc_ctx *ctx, const unsigned char *bytes, const long long *byte_off,
                            int n_utts, float *post, int *frame_off);
/* Allocates NOW every buffer a later call of up to max_rows frames in max_utts utterances would otherwise allocate on
 * demand inside its first call (device staging, pinned features / posteriors / offsets; with max_wave_bytes > 0 -- after
 * lcrc_frontend_configure -- the byte buffers of the waveform entries too).  Optional: the entry points grow their buffers
 * themselves; page pinning is what that costs (~8 ms per 32 768 HU frames), and a caller with a set-up phase calls this
 * there.  Call it after lcrc_decoder_configure / lcrc_set_posterior_readback (they decide whether a pinned posterior
 * buffer is needed at all).  Has no reference counterpart (the reference allocates per bunch, traps.cpp:63-101). */
int lcrc_reserve(lcrc_ctx *ctx, int max_rows, int max_utts, long long max_wave_bytes);
/* Zero-copy variant: lcrc_wave_stage_buffer returns the context's pinned byte buffer (valid until a later
 * call asks for more capacity); the caller reads its files straight into it -- utterance u at start[u]
 * (ascending, not overlapping, even for lin16), n_bytes[u] long -- and lcrc_wave_stage_run does what
 * lcrc_wave_to_posteriors does without copying the bytes again. */
int lcrc_wave_stage_buffer(lcrc_ctx *ctx, long long capacity, unsigned char **bytes);
int lcrc_wave_stage_run(lcrc_ctx *ctx, const long long *start, const long long *n_bytes, int n_utts,
                        float *post, int *frame_off);
/* The front-end's arithmetic up to the mel-bank ENERGIES on the GPU, everything behind them left to the caller: the
 * utterances in the wave stage buffer (as for lcrc_wave_stage_run) go through decode, window, FFT, power spectrum and the
 * bank sums -- the reference's operations in the reference's order (melbanks.cpp:111-149, dspc.cpp:24-78,236-269), so the
 * energies equal the host front-end's bit for bit -- and come back in the context's pinned feature buffer, *energies =
 * [rows][nbanks] (the buffer lcrc_stage_buffers hands out, with room for the posteriors reserved too).  The caller takes
 * ln() with ITS libm (dspc.h:155-160: x > 0 ? logf(x) : 0), applies framenorm / offlinenorm in place and calls
 * lcrc_stage_run(frame_off, n_utts): features, and therefore posteriors, identical to a host front-end's at a tenth of
 * its CPU time (the FFTs are 90 % of it).  No sentence normalisation happens here, whatever lcrc_frontend_configure said. */
int lcrc_wave_stage_energies(lcrc_ctx *ctx, const long long *start, const long long *n_bytes, int n_utts, float **energies,
                             int *frame_off);
/* post == NULL in lcrc_wave_stage_run leaves the posteriors in the context's pinned output buffer: this
 * returns it (rows as in frame_off; valid until the next call on the context) */
int lcrc_staged_posteriors(lcrc_ctx *ctx, const float **post);

/* ---- posterior writer path ("next" row f2) ---------------------------------------------
 * The softening functions SpeechRec applies to every posterior after the nets
 * (posteriors/softening_func, srec.cpp:1062-1070; decoder/softening_func, srec.cpp:1089-1097;
 * functions srec.cpp:164-176, srec.h:192-194) evaluated on the device in the merger's
 * epilogue, and optionally the byte order of HTK dumps (matrix.h:2506-2544 writes big-endian
 * floats), so that the host writes a file with one fwrite and feeds the decoder without
 * another pass.  Up to two stages are applied in order (posterior softening, then decoder
 * softening, as `-t str` does).  The setting holds for every later posterior call on this
 * context (host-pointer, staged, device-pointer, waveform and streaming forms); n_stages = 0
 * and big_endian = 0 restore plain posteriors. */
enum { LCRC_SOFT_NONE = 0, LCRC_SOFT_LOG = 1, LCRC_SOFT_IGOR = 2, LCRC_SOFT_GMM_BYPASS = 3 };
typedef struct lcrc_softening {
    int func;                /* LCRC_SOFT_*                                                               */
    float arg1, arg2, arg3;  /* igor: middle point, right log base, left log base (srec.cpp:166-171)      */
} lcrc_softening;
int lcrc_output_configure(lcrc_ctx *ctx, const lcrc_softening *stages, int n_stages, int big_endian);

/* ---- decoder on the device ("next" row f3; optional -- the shipped arrangement decodes on the host) ------
 * PhnDec (decoder/type=phndec of every shipped config; phndec.cpp:44-303): the phoneme-loop Viterbi
 * with S-state left-to-right models, ln 0.5 transitions, insertion penalty, and labels released at the
 * time_pruning horizon, run by one wave per utterance right behind the posterior kernel, on the
 * (softened: configure decoder/softening_func with lcrc_output_configure) posteriors in HBM.
 * After lcrc_decoder_configure every host-synchronous posterior call (lcrc_posteriors, _batch,
 * lcrc_stage_run, lcrc_wave_to_posteriors) also decodes; lcrc_last_labels returns the result of the most
 * recent one.  lcrc_set_posterior_readback(ctx, 0) then skips the device-to-host copy of the
 * posteriors (`post` arguments may be NULL; the staged posterior buffer is not refreshed).
 * n_phonemes <= 64, states_per_phn <= 4, time_pruning <= 63 (shipped configs: 40), n_phonemes*states <= outputs;
 * n_phonemes = 0 switches the decoder off. */
typedef struct lcrc_label {
    int start, end;          /* frames; the reference prints them as "%d00000" (100 ns units)   phndec.cpp:230 */
    int phn;                 /* line number in dicts/phoneme_list                                              */
    float score;
} lcrc_label;
int lcrc_decoder_configure(lcrc_ctx *ctx, int n_phonemes, int states_per_phn, int time_pruning, float wpenalty);
int lcrc_set_posterior_readback(lcrc_ctx *ctx, int enabled);
/* labels of utterance u: labels[first[u]] .. labels[first[u] + count[u] - 1]; valid until the next call */
int lcrc_last_labels(lcrc_ctx *ctx, const lcrc_label **labels, const int **first, const int **count, int *n_utts);
/* For callers that run list after list of staged calls (this repository's SpeechRec): the decoder of a call runs BESIDE the
 * next call's front-end and posterior kernels instead of in front of them.  With lcrc_set_decoder_overlap(ctx, 1) -- and the
 * decoder configured, read-back off -- lcrc_stage_run and lcrc_wave_stage_run(post = NULL) return as soon as their
 * posterior kernels are done; the decoder kernel follows on a second stream of the context, on the
 * call's own copies of the posteriors, offsets and label buffers (two sets alternate).  The labels of a call are then
 * fetched AFTER the next call has returned: lcrc_prev_labels waits for the decoder of the call before the most recent
 * one and returns its labels (valid until the next staged call but one); lcrc_last_labels does the same for the most
 * recent call (after the last call of a list).  Same labels as without the overlap (tested); every other entry point
 * keeps decoding synchronously.  Has no reference counterpart (the reference decodes frame by frame on the host,
 * srec.cpp:1089-1104). */
int lcrc_set_decoder_overlap(lcrc_ctx *ctx, int enabled);
/* Several contexts of one process on one device (the CLI's three per GPU): with lcrc_set_launch_order(ctx, 1) on each of
 * them, the POSTERIOR kernels of their calls run one after the other on the device, in the order the calls queued them --
 * each launch waits, on the device, for an event behind the posterior kernels of the launch queued before it --, while
 * everything else of a call (uploads, front-end kernels, decoder) still runs beside other contexts' work.  Two posterior
 * kernels that share the device finish together and later than they would one after the other; the order only removes
 * that, no result changes.  Default 0.  No reference counterpart. */
int lcrc_set_launch_order(lcrc_ctx *ctx, int ordered);
int lcrc_prev_labels(lcrc_ctx *ctx, const lcrc_label **labels, const int **first, const int **count, int *n_utts);

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
/* How the host-pointer entry points wait for the device at the end of a call.  0 (default): hipStreamSynchronize -- the
 * calling thread spins on the completion signal: lowest latency, one busy core per waiting thread.  n > 0: an event
 * behind the work is queried every n microseconds with the thread asleep in between: next to no CPU time, the completion
 * noticed up to n (plus the timer's slack) late.  For callers that keep more contexts in flight -- a thread each -- than
 * they have cores to burn: the CLI switches to it when its contexts outnumber half of the usable cores. */
int lcrc_set_wait_mode(lcrc_ctx *ctx, int poll_interval_us);
/* Optional notification for callers that keep several contexts in flight on one GPU: fn(arg) is called on the CALLING
 * thread, once per host-pointer / staged / waveform entry-point call that launches, as soon as the posterior kernel(s) of
 * that call have finished on the device -- while the call's decoder launch and copy-back may still be running -- and
 * before the call returns.  (The CLI admits a limited number of its contexts' launches to a GPU at a time; it releases a
 * launch's slot here, so that the next context's kernels start while this one's labels / posteriors travel back.)
 * fn = NULL switches it off.  Not called when a call launches nothing or fails before its launch. */
typedef void (*lcrc_kernel_done_fn)(void *arg);
int lcrc_set_kernel_done_callback(lcrc_ctx *ctx, lcrc_kernel_done_fn fn, void *arg);
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
/* Test hook, inert unless the process environment holds LCRC_FAULT_INJECTION=1 (else LCRC_E_UNSUPPORTED): the nth
 * (0 = next) staging-buffer allocation of this process from now on fails as if the device / pinned memory were
 * exhausted; -1 switches the injection off.  The failing call returns LCRC_E_NOMEM, leaves no half-allocated
 * buffer group behind, and the context stays usable. */
int lcrc_debug_fail_alloc(int nth);
/* Name of the kernel variant selected for this model ("cz_42_69_9", "generic_64_104_13", ...; the same in both arithmetics) */
const char *lcrc_kernel_name(const lcrc_ctx *ctx);

#ifdef __cplusplus
}
#endif
#endif
