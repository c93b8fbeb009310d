/* oracle/kaldi_oracle.h -- TEST INFRASTRUCTURE (see kaldi_oracle.c header: PARITY UNPINNED).
 * Prototypes mirror /root/reference/c/codee_speech.h:13-25 with an oracle_ prefix so that
 * the oracle and the product library can be loaded into one test process side by side. */
#pragma once
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

void oracle_set_fft_exact(int exact);   /* 1 (default): FFT/DCT in the wider type, rounded once; 0: working precision */
int oracle_get_fft_exact(void);

int oracle_kaldi_spectrogram_s(float *Y, float *X, const size_t N, const float sr, const float frame_length, const float frame_shift, const int snip_edges, const float dither, const int dc0, const int raw_energy, const float preemph, const char win_type[], const int mn0);
int oracle_kaldi_spectrogram_d(double *Y, double *X, const size_t N, const double sr, const double frame_length, const double frame_shift, const int snip_edges, const double dither, const int dc0, const int raw_energy, const double preemph, const char win_type[], const int mn0);

int oracle_kaldi_fbank_s(float *Y, float *X, const size_t N, const float sr, const float frame_length, const float frame_shift, const int snip_edges, const float dither, const int dc0, const int raw_energy, const float preemph, const char win_type[], const int amp, const float lof, const float hif, const size_t B, const int lg, const int use_energy, const int mn0);
int oracle_kaldi_fbank_d(double *Y, double *X, const size_t N, const double sr, const double frame_length, const double frame_shift, const int snip_edges, const double dither, const int dc0, const int raw_energy, const double preemph, const char win_type[], const int amp, const double lof, const double hif, const size_t B, const int lg, const int use_energy, const int mn0);

int oracle_kaldi_mfcc_s(float *Y, float *X, const size_t N, const float sr, const float frame_length, const float frame_shift, const int snip_edges, const float dither, const int dc0, const int raw_energy, const float preemph, const char win_type[], const float lof, const float hif, const size_t B, const size_t C, const float Q, const int use_energy, const int mn0);
int oracle_kaldi_mfcc_d(double *Y, double *X, const size_t N, const double sr, const double frame_length, const double frame_shift, const int snip_edges, const double dither, const int dc0, const int raw_energy, const double preemph, const char win_type[], const double lof, const double hif, const size_t B, const size_t C, const double Q, const int use_energy, const int mn0);

/* lpc_order == 0 selects the drop-in rule order = C-1 */
int oracle_kaldi_plp_s(float *Y, float *X, const size_t N, const float sr, const float frame_length, const float frame_shift, const int snip_edges, const float dither, const int dc0, const int raw_energy, const float preemph, const char win_type[], const float lof, const float hif, const size_t B, const float compress, const size_t C, const float Q, const float cep_scale, const int use_energy, const int mn0, const size_t lpc_order);
int oracle_kaldi_plp_d(double *Y, double *X, const size_t N, const double sr, const double frame_length, const double frame_shift, const int snip_edges, const double dither, const int dc0, const int raw_energy, const double preemph, const char win_type[], const double lof, const double hif, const size_t B, const double compress, const size_t C, const double Q, const double cep_scale, const int use_energy, const int mn0, const size_t lpc_order);

int oracle_frame_count_s(size_t N, float sr, float frame_length, float frame_shift, int snip_edges, size_t *W, size_t *L, size_t *stp);
int oracle_frame_count_d(size_t N, double sr, double frame_length, double frame_shift, int snip_edges, size_t *W, size_t *L, size_t *stp);

/* cpu_baseline.c: run one extractor over a batch of utterances with OpenMP over utterances
   (one utterance per call of the functions above, exactly as the reference CLI would).
   kind: 0 spectrogram, 1 fbank, 2 mfcc, 3 plp.  X is the packed batch, x_off/y_off element offsets.
   Returns the number of threads used, or -1 on error. */
int oracle_batch_s(int kind, size_t nutt, const float *X, const size_t *x_off, const size_t *n_samp, float *Y, const size_t *y_off,
                   float sr, float frame_length, float frame_shift, int snip_edges, int dc0, int raw_energy, float preemph, const char *win_type,
                   int amp, float lof, float hif, size_t B, int lg, size_t C, float Q, float compress, float cep_scale, int use_energy, int threads);

#ifdef __cplusplus
}
#endif
