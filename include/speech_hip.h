/* include/speech_hip.h -- batch/plan extension of the drop-in boundary (additive; C ABI).
 *
 * The reference has no handles: every call rebuilds its window, FFT plan, mel and lifter tables
 * (c/kaldi_mfcc.c:105-210) and processes ONE utterance (src/kaldi_mfcc.cpp:290).  To make
 * device-resident, ragged-batch throughput reachable the same pipeline is exposed as
 *   plan  = the per-call tables of c/kaldi_mfcc.c:105-210, built once and kept on one device;
 *   batch = the launch geometry for a ragged list of utterances (tile list, offsets);
 *   run   = the per-frame loop of c/kaldi_mfcc.c:213-334 for every utterance of the batch.
 * Plain pointers and sizes only; device pointers are what hipMalloc (or torch) returns and the
 * stream is a hipStream_t passed as void*.  All functions return 0 on success, 1 on error
 * (message via speech_last_error(), also printed to stderr in the reference's style).
 * There is no CPU fallback anywhere behind this header.
 */
#pragma once
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { SPEECH_SPECTROGRAM = 0, SPEECH_FBANK = 1, SPEECH_MFCC = 2, SPEECH_PLP = 3 };
enum { SPEECH_F32 = 0, SPEECH_F64 = 1 };

/* The union of the reference functions' option lists (c/codee_speech.h:13-25). */
typedef struct speech_opts {
    int kind;               /* SPEECH_* : which reference function family */
    int dtype;              /* SPEECH_F32 = the *_s functions, SPEECH_F64 = the *_d functions */
    double sr, frame_length, frame_shift;
    int snip_edges;
    double dither;          /* >FLT_EPSILON: seeded counter-based Gaussian dither (not bit-equal to the
                               reference's wall-clock-seeded PCG, c/kaldi_mfcc.c:87-98) */
    int dc0, raw_energy;
    double preemph;
    char win_type[32];
    int amp;                /* fbank */
    double lof, hif;        /* fbank, mfcc, plp */
    size_t B;
    int lg;                 /* fbank */
    int use_energy;
    size_t C;               /* mfcc, plp */
    double Q;
    double compress, cep_scale;   /* plp */
    size_t lpc_order;       /* plp; 0 = C-1 (the drop-in rule) */
    int mn0;
    unsigned long long dither_seed;
    int force_generic;      /* 1 = use the generic (any-size, any-precision) kernels even where a tuned
                               gfx950 kernel exists; for testing */
} speech_opts;

typedef struct speech_plan speech_plan;
typedef struct speech_batch speech_batch;

/* Fill opts with the reference CLI's defaults for `kind` (srci/kaldi_mfcc.cpp:106-172), dither=0. */
void speech_opts_default(speech_opts *opts, int kind);

int speech_device_count(void);
const char *speech_last_error(void);

/* Validate opts exactly as the reference prologue does (c/kaldi_mfcc.c:54-78) and build the tables on `device`. */
int speech_plan_create(speech_plan **plan, const speech_opts *opts, int device);
void speech_plan_destroy(speech_plan *plan);
/* L, stp, nfft, feature dimension D, and in *tuned which frame kernel serves this plan: 1 = the tuned float kernels, 2 = the
 * 4-frames-per-wave double kernel (512-point frames without dither), 0 = the generic kernel. Any pointer may be NULL. */
int speech_plan_sizes(const speech_plan *plan, size_t *L, size_t *stp, size_t *nfft, size_t *D, int *tuned);
/* W for an utterance of N samples; returns 1 (and *W=0) if the reference would reject N (N<1, or snip_edges && L>N). */
int speech_plan_num_frames(const speech_plan *plan, size_t N, size_t *W);

/* Geometry for nutt utterances.  x_off[u]/y_off[u] are ELEMENT offsets of utterance u inside the packed input /
   output device arrays; pass NULL for dense packing in order (x_off[u] = sum of n_samples before u, y_off likewise
   with W*D).  Utterances the reference would reject make this call fail. */
int speech_batch_create(speech_batch **batch, speech_plan *plan, size_t nutt, const size_t *n_samples,
                        const size_t *x_off, const size_t *y_off);
void speech_batch_destroy(speech_batch *batch);
/* totals: samples read, features written (elements), frames; and the element offsets actually used. */
int speech_batch_totals(const speech_batch *batch, size_t *total_samples, size_t *total_out, size_t *total_frames);
int speech_batch_offsets(const speech_batch *batch, size_t *x_off, size_t *y_off, size_t *n_frames);
/* Dither only: utterance u of the batch draws its noise as utterance number ids[u] (default, or ids == NULL: u), so that a
   corpus cut into several batches, or sharded over ranks, gets the noise it would get as one batch.  Waits for the device. */
int speech_batch_set_ids(speech_batch *batch, const unsigned int *ids);

/* Enqueue the whole batch on `stream` (hipStream_t as void*, NULL = default stream).  X and Y are device
   pointers of the plan's dtype.  Asynchronous: returns after the launches are queued. */
int speech_batch_run(speech_batch *batch, const void *X_dev, void *Y_dev, void *stream);
/* Same, but brackets the dominant (per-frame pipeline) kernel with HIP events recorded on `stream`, waits for
   completion and returns that kernel's duration in milliseconds. */
int speech_batch_run_timed(speech_batch *batch, const void *X_dev, void *Y_dev, void *stream, float *kernel_ms);

/* Widening (no counterpart in the reference, whose tools take float tensors only, srci/kaldi_mfcc.cpp:6):
   the same batch with the audio as 16-bit PCM on the device (same element offsets).  Samples are converted to the plan's
   dtype as they are loaded (exact), so the features equal speech_batch_run on the converted audio bit for bit, for half the
   input bytes over PCIe and HBM.  kernel_ms may be NULL (asynchronous) or receives the per-frame kernel's duration. */
int speech_batch_run_pcm16(speech_batch *batch, const short *X16_dev, void *Y_dev, void *stream, float *kernel_ms);

/* The same widening for host callers: one utterance of 16-bit PCM in host memory -> features in host memory (opts->dtype),
   exactly as a drop-in call of include/codee_speech.h with these options would return them for the converted audio
   (validation, error style, plan cache, staging buffers), with half the bytes over PCIe.  Y must hold W x D elements
   (speech_plan_num_frames / the reference's W formula, c/kaldi_mfcc.c:77). */
int speech_extract_pcm16(const speech_opts *opts, const short *X16_host, size_t N, void *Y_host);

/* A ragged batch from host memory to host memory: what the reference's callers hold (src/kaldi_mfcc.cpp:286-308), for many
   utterances at once.  X[u] points at utterance u's n_samples[u] samples (the plan's dtype, or 16-bit PCM when pcm16 != 0),
   Y[u] at room for its W_u x D features (the plan's dtype).  The utterances are cut into chunks of about chunk_bytes of
   input (0 = 64 MiB; a chunk holds at least one utterance); each chunk is packed into pinned memory, copied, run as one
   batch and copied back on one of two streams, so that copies and packing overlap the kernels.  The features equal
   speech_batch_run on the whole list bit for bit (dithered plans included; ids, if not NULL, gives the utterances' numbers
   for the dither generator as speech_batch_set_ids does -- a rank that holds a shard of a corpus passes the corpus
   indices and gets the noise of the unsharded run).  Synchronous: Y is complete on return.  The
   staging buffers (device and pinned, about 2 x (chunk input + chunk output)) belong to the calling thread and are kept
   for later calls; speech_host_release frees them. */
int speech_plan_run_host(speech_plan *plan, size_t nutt, const void *const *X, const size_t *n_samples, void *const *Y,
                         int pcm16, size_t chunk_bytes, const unsigned int *ids);
void speech_host_release(void);

/* The same from ONE call over several GPUs of a node: what a C caller of the reference (src/kaldi_mfcc.cpp:283-308, one call
   per utterance on one thread) needs to use all of them.  devices[0 .. ndev-1] are HIP device numbers (a number may repeat:
   two pipelines on one device).  The utterances are dealt to the devices by speech_shard_utterances (longest first to the
   least loaded device); every device gets its own host thread, its own plan (tables on that device) and its
   own two-stream pinned staging pipeline, exactly speech_plan_run_host's; there is no collective and no peer copy.  Utterance
   u draws its dither as utterance number ids[u] (NULL: u), so the features equal the single-device run bit for bit whatever
   ndev is.  Synchronous.  The option set and every utterance's length are validated before any device is touched (messages
   as the drop-in functions').  A device that fails (cannot be selected, an allocation or any HIP call fails) does not fail
   the call while another device works: its utterances are dealt again, once, to the devices that finished their share, with
   a line on stderr; the call returns 1 only if every device failed or the second deal fails too. */
int speech_run_host_multi(const speech_opts *opts, int ndev, const int *devices, size_t nutt, const void *const *X,
                          const size_t *n_samples, void *const *Y, int pcm16, size_t chunk_bytes, const unsigned int *ids);
/* The assignment alone (host arithmetic, no device): shard[u] in [0, nshards) for every utterance; longest-processing-time
   greedy on the sample counts, ties to the lower index / lower shard.  Identical to speech_amd.api.shard_utterances. */
int speech_shard_utterances(size_t nutt, const size_t *n_samples, int nshards, int *shard);
/* Validation and sizes without a plan or a device: checks opts as the reference prologue does (c/kaldi_mfcc.c:54-78) and N as
   c/kaldi_mfcc.c:70-77, and returns the frame count W of an utterance of N samples and the feature dimension D (either
   pointer may be NULL) -- what a caller needs to size Y, as src/kaldi_mfcc.cpp:255-261 computes it by hand. */
int speech_opts_frames(const speech_opts *opts, size_t N, size_t *W, size_t *D);

/* Widening: corpus post-processing on a batch's packed features Y_dev (frame-major W x D per utterance,
   the layout speech_batch_run writes).  The reference's only post-pass is mn0, the per-utterance mean subtraction
   (c/kaldi_fbank.c:355-370, c/kaldi_mfcc.c:337-352); these generalise it to Kaldi's apply-cmvn, add-deltas and
   splice-feats.  All enqueue on `stream`; speech_batch_stats also waits for it.  Not thread-safe per batch (they share
   the batch's moment buffer).
   - stats: per-feature sum and sum of squares over every frame of the batch, in double, into host arrays of D; *frames =
     number of frames.  A multi-GPU caller adds these over ranks (the one place this path has a collective: 2 D + 1 doubles)
     and applies the result with speech_batch_affine.
   - cmvn: per utterance, in place: subtract the mean; norm_vars != 0 also divides by the standard deviation (variance
     floored at 1e-20).  norm_vars == 0 is mn0 on D columns.
   - affine: in place y = y * scale[d] + offset[d] (multiply, then add, each rounded in the plan's dtype); scale / offset
     are host arrays of D doubles.
   - add_deltas: Z[t] = [Y[t] | delta | delta-delta | ...], D (order + 1) columns, order in [0 3], window in [1 5], filters
     and edge clamping as Kaldi's DeltaFeatures; utterance u of Z starts at (order + 1) x its offset in Y.
   - splice: Z[t] = [Y[t-left] | ... | Y[t+right]] with the frame index clamped to the utterance, D (left + right + 1)
     columns; utterance u of Z starts at (left + right + 1) x its offset in Y.  Z must not alias Y.
   - splice_subsample: splice-feats followed by subsample-feats --n=n --offset=offset in one pass (left = right = 0: the
     subsampling alone): row j of an utterance's output is its spliced frame offset + j n, ceil((W - offset) / n) rows (none
     if W <= offset); n >= 1, 0 <= offset < n.  The utterances' outputs are packed back to back in batch order;
     speech_batch_subsample_rows gives the rows per utterance (rows may be NULL) and their total.  The first call with a
     new (n, offset) waits for the stream and uploads a table of nutt offsets; later calls only launch. */
int speech_batch_stats(speech_batch *batch, const void *Y_dev, double *sum, double *sumsq, unsigned long long *frames, void *stream);
int speech_batch_cmvn(speech_batch *batch, void *Y_dev, int norm_vars, void *stream);
int speech_batch_affine(speech_batch *batch, void *Y_dev, const double *scale, const double *offset, void *stream);
int speech_batch_add_deltas(speech_batch *batch, const void *Y_dev, void *Z_dev, int order, int window, void *stream);
int speech_batch_splice(speech_batch *batch, const void *Y_dev, void *Z_dev, int left, int right, void *stream);
int speech_batch_splice_subsample(speech_batch *batch, const void *Y_dev, void *Z_dev, int left, int right, int n, int offset, void *stream);
int speech_batch_subsample_rows(const speech_batch *batch, int n, int offset, size_t *rows, size_t *total_rows);

/* speech_batch_cmvn for callers that cannot allocate: code under HIP graph capture, tight serving loops, harnesses that
   forbid hipMalloc.  No counterpart in the reference (its mn0 pass, c/kaldi_fbank.c:355-370, works in the caller's Y).
   Same arguments and the same result, bit for bit, as speech_batch_cmvn; the per-utterance moments ([nutt][2][D] doubles)
   live in a static device array of 4 MiB per GPU instead of a buffer the batch allocates on first use, and the call makes
   no hipMalloc / hipFree and no other allocating call.  Batches that need more (nutt x 2 x D > 524 288) are rejected with a
   message: use speech_batch_cmvn.  Every call on a GPU shares that GPU's array: calls that overlap on one GPU (two streams,
   two threads) race; calls on one stream are ordered and safe.  Nothing in the array has to start at zero. */
int speech_batch_cmvn_noalloc(speech_batch *batch, void *Y_dev, int norm_vars, void *stream);

/* Dither of the ten drop-in functions (include/codee_speech.h).  The reference seeds its generator from the wall clock on
   every call (c/kaldi_fbank.c:97-101), so no two calls add the same noise; here a dithered drop-in call draws the noise of
   utterance number k of a counter-based generator under `seed`, k = 0, 1, 2, ... counting the dithered drop-in calls of the
   process (default seed 0x5EED2026).  utt >= 0 pins k = utt for every following call (reproducible runs; a caller that
   extracts a corpus one call per utterance passes the utterance's number before each call); utt < 0 returns to the counter
   and restarts it at 0.  Process-wide; the batch ABI is not affected (speech_opts.dither_seed, speech_batch_set_ids). */
void speech_set_dropin_dither(unsigned long long seed, long long utt);

/* Bench / test support (no counterpart in the reference): fill the batch's packed audio X_dev (the plan's dtype, or int16
   when pcm16 != 0) with the synthetic corpus of SURVEY.md 8(d) -- an integer-only counter-based generator (near-Gaussian
   noise at about -29 dBFS plus three triangle tones per utterance, int16 scale), so that host (speech_amd/corpus.py) and
   device produce the same samples bit for bit and a 10k-hour corpus never has to exist on the host.  Utterance u of the
   batch is corpus utterance ids[u] (host array of nutt; NULL: u).  Enqueues on `stream` (waits for it first when ids is
   given).  The batch must use its default dense packing or offsets that do not overlap. */
int speech_batch_synth_audio(speech_batch *batch, void *X_dev, int pcm16, unsigned long long seed, const unsigned long long *ids, void *stream);
/* The same generator on the host (no device needed): samples 0 .. n-1 of corpus utterance `utt` as int16. */
int speech_synth_audio_host(unsigned long long seed, unsigned long long utt, size_t n, short *out);

#ifdef __cplusplus
}
#endif
