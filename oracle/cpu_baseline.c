/* oracle/cpu_baseline.c -- TEST/BENCH INFRASTRUCTURE (see kaldi_oracle.c header).
 * The CPU leg that bench.py times beside the GPU: one reference-style call per utterance
 * (tables rebuilt per call, per-frame loop, sparse mel -- as c/kaldi_mfcc.c:105-334 does),
 * parallelised over utterances with OpenMP because utterances are independent. */
#include <stdio.h>
#include <stddef.h>
#ifdef _OPENMP
#include <omp.h>
#endif
#include "kaldi_oracle.h"

int oracle_batch_s(int kind, size_t nutt, const float *X, const size_t *x_off, const size_t *n_samp, float *Y, const size_t *y_off,
                   float sr, float frame_length, float frame_shift, int snip_edges, int dc0, int raw_energy, float preemph, const char *win_type,
                   int amp, float lof, float hif, size_t B, int lg, size_t C, float Q, float compress, float cep_scale, int use_energy, int threads)
{
    int used = 1, err = 0;
#ifdef _OPENMP
    if (threads > 0) { omp_set_num_threads(threads); }
    used = omp_get_max_threads();
#else
    (void)threads;
#endif
    #pragma omp parallel for schedule(dynamic, 1) reduction(|:err)
    for (long long u = 0; u < (long long)nutt; ++u)
    {
        float *y = Y + y_off[u];
        float *x = (float *)(X + x_off[u]);
        int rc;
        switch (kind)
        {
            case 0: rc = oracle_kaldi_spectrogram_s(y, x, n_samp[u], sr, frame_length, frame_shift, snip_edges, 0.0f, dc0, raw_energy, preemph, win_type, 0); break;
            case 1: rc = oracle_kaldi_fbank_s(y, x, n_samp[u], sr, frame_length, frame_shift, snip_edges, 0.0f, dc0, raw_energy, preemph, win_type, amp, lof, hif, B, lg, use_energy, 0); break;
            case 2: rc = oracle_kaldi_mfcc_s(y, x, n_samp[u], sr, frame_length, frame_shift, snip_edges, 0.0f, dc0, raw_energy, preemph, win_type, lof, hif, B, C, Q, use_energy, 0); break;
            case 3: rc = oracle_kaldi_plp_s(y, x, n_samp[u], sr, frame_length, frame_shift, snip_edges, 0.0f, dc0, raw_energy, preemph, win_type, lof, hif, B, compress, C, Q, cep_scale, use_energy, 0, 0u); break;
            default: rc = 1;
        }
        err |= rc;
    }
    return err ? -1 : used;
}
