/* include/codee_speech.h -- the drop-in boundary.
 *
 * The ten exported functions of the reference's BLAS-like C ABI, with the reference's own
 * argument lists (reference: c/codee_speech.h:13-25), implemented by libspeech.so on an
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

#ifdef __cplusplus
namespace codee {
extern "C" {
#endif

/* replaces c/kaldi_fbank.c:59 (kaldi_fbank_s) and :380 (kaldi_fbank_d) */
int kaldi_fbank_s (float *Y, float *X, const size_t N, const float sr, const float frame_length, const float frame_shift, const int snip_edges, const float dither, const int dc0, const int raw_energy, const float preemph, const char win_type[], const int amp, const float lof, const float hif, const size_t B, const int lg, const int use_energy, const int mn0);
int kaldi_fbank_d (double *Y, double *X, const size_t N, const double sr, const double frame_length, const double frame_shift, const int snip_edges, const double dither, const int dc0, const int raw_energy, const double preemph, const char win_type[], const int amp, const double lof, const double hif, const size_t B, const int lg, const int use_energy, const int mn0);
/* replaces c/kaldi_fbank.c:668 and :920 (hard-wired sr=16000, frame_length=25 ms, B=80; see DESIGN.md "fbank_default") */
int kaldi_fbank_default_s (float *Y, float *X, const size_t N, const float frame_shift, const int snip_edges, const float dither, const int dc0, const int raw_energy, const float preemph, const char win_type[], const int amp, const float lof, const float hif, const int lg, const int use_energy, const int mn0);
int kaldi_fbank_default_d (double *Y, double *X, const size_t N, const double frame_shift, const int snip_edges, const double dither, const int dc0, const int raw_energy, const double preemph, const char win_type[], const int amp, const double lof, const double hif, const int lg, const int use_energy, const int mn0);

/* replaces c/kaldi_mfcc.c:52 (kaldi_mfcc_s) and :363 (kaldi_mfcc_d) */
int kaldi_mfcc_s (float *Y, float *X, const size_t N, const float sr, const float frame_length, const float frame_shift, const int snip_edges, const float dither, const int dc0, const int raw_energy, const float preemph, const char win_type[], const float lof, const float hif, const size_t B, const size_t C, const float Q, const int use_energy, const int mn0);
int kaldi_mfcc_d (double *Y, double *X, const size_t N, const double sr, const double frame_length, const double frame_shift, const int snip_edges, const double dither, const int dc0, const int raw_energy, const double preemph, const char win_type[], const double lof, const double hif, const size_t B, const size_t C, const double Q, const int use_energy, const int mn0);

/* replaces c/kaldi_plp.c:53 (kaldi_plp_s; does not compile in the reference) and :434 (kaldi_plp_d; commented out).
   LPC order = C-1 (the ABI has no order argument); speech_hip.h exposes it explicitly. */
int kaldi_plp_s (float *Y, float *X, const size_t N, const float sr, const float frame_length, const float frame_shift, const int snip_edges, const float dither, const int dc0, const int raw_energy, const float preemph, const char win_type[], const float lof, const float hif, const size_t B, const float compress, const size_t C, const float Q, const float cep_scale, const int use_energy, const int mn0);
int kaldi_plp_d (double *Y, double *X, const size_t N, const double sr, const double frame_length, const double frame_shift, const int snip_edges, const double dither, const int dc0, const int raw_energy, const double preemph, const char win_type[], const double lof, const double hif, const size_t B, const double compress, const size_t C, const double Q, const double cep_scale, const int use_energy, const int mn0);

/* replaces c/kaldi_spectrogram.c:46 (kaldi_spectrogram_s) and :281 (kaldi_spectrogram_d) */
int kaldi_spectrogram_s (float *Y, float *X, const size_t N, const float sr, const float frame_length, const float frame_shift, const int snip_edges, const float dither, const int dc0, const int raw_energy, const float preemph, const char win_type[], const int mn0);
int kaldi_spectrogram_d (double *Y, double *X, const size_t N, const double sr, const double frame_length, const double frame_shift, const int snip_edges, const double dither, const int dc0, const int raw_energy, const double preemph, const char win_type[], const int mn0);

#ifdef __cplusplus
}
}
#endif
