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

/* the arguments every extractor starts with, in the order of the reference's ABI (see include/codee_speech.h) */
#define ORACLE_FRONT(T) \
    T *Y, T *X, size_t N, T sr, T frame_length, T frame_shift, int snip_edges, T dither, int dc0, int raw_energy, T preemph, const char win_type[]
#define ORACLE_MEL(T) T lof, T hif, size_t B

int oracle_kaldi_spectrogram_s(ORACLE_FRONT(float), int mn0);
int oracle_kaldi_spectrogram_d(ORACLE_FRONT(double), int mn0);

int oracle_kaldi_fbank_s(ORACLE_FRONT(float), int amp, ORACLE_MEL(float), int lg, int use_energy, int mn0);
int oracle_kaldi_fbank_d(ORACLE_FRONT(double), int amp, ORACLE_MEL(double), int lg, int use_energy, int mn0);

int oracle_kaldi_mfcc_s(ORACLE_FRONT(float), ORACLE_MEL(float), size_t C, float Q, int use_energy, int mn0);
int oracle_kaldi_mfcc_d(ORACLE_FRONT(double), ORACLE_MEL(double), size_t C, double Q, int use_energy, int mn0);

/* lpc_order == 0 selects the drop-in rule order = C-1 */
int oracle_kaldi_plp_s(ORACLE_FRONT(float), ORACLE_MEL(float), float compress, size_t C, float Q, float cep_scale, int use_energy, int mn0, size_t lpc_order);
int oracle_kaldi_plp_d(ORACLE_FRONT(double), ORACLE_MEL(double), double compress, size_t C, double Q, double cep_scale, int use_energy, int mn0, size_t lpc_order);

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
