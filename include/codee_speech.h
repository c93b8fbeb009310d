/* include/codee_speech.h -- the drop-in boundary.
 *
 * The ten exported functions of the reference's BLAS-like C ABI, with the reference's own
 * argument lists (reference: c/codee_speech.h:13-25; top-level const on by-value parameters is
 * not part of a C prototype and is left out), implemented by libspeech.so on an
 * MI355X (gfx950) through HIP kernels.  The header is self-contained: the reference's copy
 * includes three sibling-repo headers (c/codee_speech.h:3-5) that callers of this ABI never
 * needed.  C++ callers see the functions in namespace codee, as in the reference
 * (c/codee_speech.h:8-11); the symbols are unmangled C symbols either way.
 *
 * Contract (reference file:line in brackets):
 *   - X[N] and Y[W*D] are HOST pointers owned by the caller; X is read only; Y must be sized
 *     by the caller from W = snip_edges ? 1+(N-L)/stp : (N+stp/2)/stp  [c/kaldi_mfcc.c:68-78],
 *     D = nfft/2+1 (spectrogram), B(+1 with use_energy) (fbank), C (mfcc, plp); frame-major.
 *   - returns 0 on success; on invalid arguments prints "error in <fn>: ..." to stderr and
 *     returns 1 [c/kaldi_mfcc.c:54-78]; W==0 returns 0 without touching Y [c/kaldi_mfcc.c:78].
 *   - there is NO CPU fallback: if no HIP device is usable the call prints an error and returns 1.
 *   - calls are synchronous and thread-safe.
 * Each call stages X to the current HIP device, runs the kernels and copies Y back; callers that
 * need throughput keep data resident and use the batch ABI in speech_hip.h instead.
 */
#pragma once
#include <stddef.h>

/* The arguments every function starts with (T = float for *_s, double for *_d), in the reference's order:
 *   Y, X            output features (W x D, frame-major) and input audio (N samples), host memory
 *   N               samples in X
 *   sr              sample rate in Hz
 *   frame_length    in ms;  L  = (size_t)(sr * frame_length / 1000) samples, evaluated in T
 *   frame_shift     in ms; stp = (size_t)(sr * frame_shift  / 1000) samples
 *   snip_edges      1: only whole frames inside X; 0: frames centred on multiples of stp, reflected at the ends
 *   dither          standard deviation of the Gaussian noise added to every frame (0 = off)
 *   dc0             subtract the frame mean
 *   raw_energy      frame energy before (1) or after (0) pre-emphasis and window
 *   preemph         pre-emphasis coefficient (0 = off)
 *   win_type        "rectangular" | "blackman" | "hamming" | "hann" | "povey" (prefix match)
 * and the one every function ends with:
 *   mn0             subtract the per-feature mean over the utterance's frames
 */
#define CODEE_SPEECH_FRONT(T) \
    T *Y, T *X, size_t N, T sr, T frame_length, T frame_shift, int snip_edges, T dither, int dc0, int raw_energy, T preemph, const char win_type[]

/* Mel filterbank arguments: lowest / highest frequency of the bank in Hz and the number of mel bins. */
#define CODEE_SPEECH_MEL(T) T lof, T hif, size_t B

#ifdef __cplusplus
namespace codee {
extern "C" {
#endif

/* kaldi_spectrogram: log power spectrum, D = nfft / 2 + 1.
   Replaces c/kaldi_spectrogram.c:46 (kaldi_spectrogram_s) and :281 (kaldi_spectrogram_d). */
int kaldi_spectrogram_s(CODEE_SPEECH_FRONT(float), int mn0);
int kaldi_spectrogram_d(CODEE_SPEECH_FRONT(double), int mn0);

/* kaldi_fbank: mel filterbank energies, D = B (+ 1 with use_energy: the log frame energy comes first).
   amp: filter the amplitude instead of the power spectrum; lg: take the log.
   Replaces c/kaldi_fbank.c:59 (kaldi_fbank_s) and :380 (kaldi_fbank_d). */
int kaldi_fbank_s(CODEE_SPEECH_FRONT(float), int amp, CODEE_SPEECH_MEL(float), int lg, int use_energy, int mn0);
int kaldi_fbank_d(CODEE_SPEECH_FRONT(double), int amp, CODEE_SPEECH_MEL(double), int lg, int use_energy, int mn0);

/* kaldi_fbank_default: kaldi_fbank with sr = 16000, frame_length = 25 ms and B = 80 hard-wired, so those three arguments
   are absent.  Replaces c/kaldi_fbank.c:668 and :920. */
int kaldi_fbank_default_s(float *Y, float *X, size_t N, float frame_shift, int snip_edges, float dither, int dc0, int raw_energy, float preemph,
                          const char win_type[], int amp, float lof, float hif, int lg, int use_energy, int mn0);
int kaldi_fbank_default_d(double *Y, double *X, size_t N, double frame_shift, int snip_edges, double dither, int dc0, int raw_energy, double preemph,
                          const char win_type[], int amp, double lof, double hif, int lg, int use_energy, int mn0);

/* kaldi_mfcc: C cepstral coefficients (C <= B) of the log mel energies, liftered with Q (0 = no lifter); use_energy puts
   the log frame energy in place of the 0th coefficient.  D = C.
   Replaces c/kaldi_mfcc.c:52 (kaldi_mfcc_s) and :363 (kaldi_mfcc_d). */
int kaldi_mfcc_s(CODEE_SPEECH_FRONT(float), CODEE_SPEECH_MEL(float), size_t C, float Q, int use_energy, int mn0);
int kaldi_mfcc_d(CODEE_SPEECH_FRONT(double), CODEE_SPEECH_MEL(double), size_t C, double Q, int use_energy, int mn0);

/* kaldi_plp: perceptual linear prediction cepstra; compress = exponent of the amplitude compression (1/3), cep_scale = final
   scale.  LPC order = C - 1 (the ABI has no order argument; speech_hip.h exposes it).  D = C.
   Replaces c/kaldi_plp.c:53 (kaldi_plp_s, which does not compile in the reference) and :434 (kaldi_plp_d, commented out). */
int kaldi_plp_s(CODEE_SPEECH_FRONT(float), CODEE_SPEECH_MEL(float), float compress, size_t C, float Q, float cep_scale, int use_energy, int mn0);
int kaldi_plp_d(CODEE_SPEECH_FRONT(double), CODEE_SPEECH_MEL(double), double compress, size_t C, double Q, double cep_scale, int use_energy, int mn0);

#ifdef __cplusplus
}
}
#endif
