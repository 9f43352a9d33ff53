/*
 * lcrc_pipeline.h -- the entry points of libphnrec_lcrc.so beyond the Traps seam (lcrc.h):
 *   - the rows SURVEY 8 marks "next": the mel-bank front-end on the GPU (f1), the posterior writer path (f2), the
 *     PhnDec decoder on the device (f3);
 *   - what a host that drives lists through several contexts per GPU uses (this repository's SpeechRec / phnrec CLI):
 *     zero-copy staging, buffers reserved ahead, device warm-up, launch order, decoder overlap, completion callback.
 * Same library, same conventions as lcrc.h.  None of it is needed to put the library behind the reference's Traps.
 */
#ifndef PHNREC_LCRC_PIPELINE_H
#define PHNREC_LCRC_PIPELINE_H

#include "lcrc.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- device start-up ------------------------------------------------------------------------ */
/* Optional: starts the HIP runtime and GPU `device_id`'s context (~0.2 s in a fresh process, by far the largest part of
 * a first lcrc_create) and returns when they are up.  Thread-safe; meant to be called from a helper thread at program
 * start so that the caller's own initialisation -- configuration, the model files and their re-packing inside
 * lcrc_create, the first file's front-end -- overlaps with it (the CLI does, one thread per device it will use).  It also
 * brings the posterior kernels' code object onto the device (20-40 ms that a
 * context's creation or first launch pays otherwise; beside the caller's
 * lcrc_create -- stream, weight upload -- it costs nothing: a one-file `phnrec` run 0.15-0.24 s instead of 0.21-0.29 on
 * the same boxes).  No reference counterpart: the reference has no device to bring up. */
int lcrc_device_warmup(int device_id);
/* The same for the other code objects of the library: the GPU front-end's and the device decoder's kernels (15-20 ms each that
 * the first waveform / decoder launch of the process pays otherwise: a list's first launch 0.02-0.05 -> 0.003 s).  For the
 * helper thread that ran lcrc_device_warmup, when the run will use them -- a one-file run that does not is 10 ms sooner done
 * without (the loads compete with its one context's creation). */
enum { LCRC_PRELOAD_FRONTEND = 1, LCRC_PRELOAD_DECODER = 2 };
int lcrc_device_preload(int device_id, int what);
/* PCI address of GPU `device_id` ("0000:c1:00.0") into buf: lets a host that drives several GPUs place the threads of
 * each near it (the CLI pins a GPU's worker threads to the CPUs of /sys/bus/pci/devices/<id>/numa_node).  No reference
 * counterpart. */
int lcrc_device_pci_bus_id(int device_id, char *buf, int len);

/* ---- zero-copy staging ------------------------------------------------------------------------ */
/* Zero-copy variant of lcrc_posteriors_batch for callers that assemble batches themselves
 * (this repository's SpeechRec does): lcrc_stage_buffers returns pinned host buffers owned
 * by the context with room for `rows` frames (valid until the next lcrc_stage_buffers call
 * with a larger size, or lcrc_destroy); the caller writes mel[rows][nbanks] into *mel,
 * calls lcrc_stage_run (synchronous: the kernel reads the features where they lie and stores
 * the posteriors straight into *post, both over PCIe while it runs -- no copy commands) and
 * reads post[rows][n_out] from *post.  `rows` of lcrc_stage_run is off[n_utts]. */
int lcrc_stage_buffers(lcrc_ctx *ctx, int rows, float **mel, float **post);
int lcrc_stage_run(lcrc_ctx *ctx, const int *off, int n_utts);

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


/* ---- several contexts in flight --------------------------------------------------------------- */
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

#ifdef __cplusplus
}
#endif
#endif
