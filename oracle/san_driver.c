/* oracle/san_driver.c -- TEST INFRASTRUCTURE: runs the C oracle over its edge cases under AddressSanitizer and
 * UndefinedBehaviorSanitizer (make -C oracle san; tests/test_oracle.py).  The reference has real out-of-bounds hazards at
 * these edges (F2B sized B*(F/2), C == 1 underflowing C-2, tiny frames); the oracle defines behaviour
 * for them and this driver checks that it does so without touching memory it does not own. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include "kaldi_oracle.h"

static int fails = 0;
#define EXPECT(c) do { if (!(c)) { fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #c); ++fails; } } while (0)

int main(void)
{
    enum { NMAX = 48000 };
    float *x = malloc(NMAX * sizeof(float));
    double *xd = malloc(NMAX * sizeof(double));
    for (int i = 0; i < NMAX; ++i) { x[i] = 1000.0f * sinf(0.013f * (float)i) + (float)((i * 2654435761u) % 2001u) - 1000.0f; xd[i] = (double)x[i]; }
    const char *wins[] = {"rectangular", "blackman", "hamming", "hann", "povey"};
    const size_t lens[] = {1, 2, 3, 159, 160, 399, 400, 401, 559, 560, 561, 2000};
    for (size_t li = 0; li < sizeof(lens) / sizeof(lens[0]); ++li)
    {
        const size_t N = lens[li];
        for (int snip = 0; snip <= 1; ++snip)
        {
            for (int wi = 0; wi < 5; ++wi)
            {
                size_t W = 0, L = 0, stp = 0;
                const int rcw = oracle_frame_count_s(N, 16000.0f, 25.0f, 10.0f, snip, &W, &L, &stp);
                /* exactly W x D elements, so that a write past the last frame is caught */
                float *y = malloc((W * 257 + 1) * sizeof(float));
                double *yd = malloc((W * 257 + 1) * sizeof(double));
                const int rc = oracle_kaldi_spectrogram_s(y, x, N, 16000.0f, 25.0f, 10.0f, snip, 0.0f, wi & 1, (wi >> 1) & 1, 0.97f, wins[wi], li & 1);
                EXPECT(rc == rcw || W == 0);
                (void)oracle_kaldi_spectrogram_d(yd, xd, N, 16000.0, 25.0, 10.0, snip, 0.0, 1, 1, 0.97, wins[wi], 0);
                free(y); free(yd);
                const size_t Bs[] = {1, 2, 3, 13, 23, 40, 80, 128, 200};
                for (size_t bi = 0; bi < sizeof(Bs) / sizeof(Bs[0]); ++bi)
                {
                    const size_t B = Bs[bi];
                    float *yf = malloc((W * (B + 1) + 1) * sizeof(float));
                    (void)oracle_kaldi_fbank_s(yf, x, N, 16000.0f, 25.0f, 10.0f, snip, 0.0f, 1, 1, 0.97f, wins[wi], (int)(bi & 1), 20.0f, 8000.0f, B, 1, 1, (int)(bi & 1));
                    (void)oracle_kaldi_fbank_s(yf, x, N, 16000.0f, 25.0f, 10.0f, snip, 0.0f, 0, 0, 0.0f, wins[wi], 0, 0.0f, 9000.0f, B, 0, 0, 0);      /* hif above Nyquist */
                    free(yf);
                    const size_t Cs[] = {1, 2, B < 13 ? B : 13, B};
                    for (int ci = 0; ci < 4; ++ci)
                    {
                        const size_t C = Cs[ci] < 1 ? 1 : Cs[ci];
                        if (C > B) { continue; }
                        float *ym = malloc((W * C + 1) * sizeof(float));
                        double *ymd = malloc((W * C + 1) * sizeof(double));
                        (void)oracle_kaldi_mfcc_s(ym, x, N, 16000.0f, 25.0f, 10.0f, snip, 0.0f, 1, 1, 0.97f, wins[wi], 20.0f, 8000.0f, B, C, (ci & 1) ? 0.0f : 22.0f, ci & 1, ci & 1);
                        (void)oracle_kaldi_mfcc_d(ymd, xd, N, 16000.0, 25.0, 10.0, snip, 0.0, 1, 0, 0.97, wins[wi], 20.0, 8000.0, B, C, 22.0, 0, 0);
                        (void)oracle_kaldi_plp_s(ym, x, N, 16000.0f, 25.0f, 10.0f, snip, 0.0f, 1, 1, 0.97f, wins[wi], 20.0f, 8000.0f, B, 1.0f / 3.0f, C, 22.0f, 1.0f, ci & 1, 0, 0);
                        (void)oracle_kaldi_plp_d(ymd, xd, N, 16000.0, 25.0, 10.0, snip, 0.0, 1, 1, 0.97, wins[wi], 20.0, 8000.0, B, 0.5, C, 22.0, 1.0, 0, 0, C > 2 ? 2 : 0);
                        free(ym); free(ymd);
                    }
                }
            }
        }
    }
    /* other frame geometries: tiny frames (nfft 1, 2, 4), a 7-sample shift, long frames */
    const float fls[] = {0.0625f, 0.125f, 0.25f, 1.0f, 20.0f, 32.0f, 64.0f, 100.0f};
    for (size_t fi = 0; fi < sizeof(fls) / sizeof(fls[0]); ++fi)
    {
        size_t W = 0, L = 0, stp = 0;
        (void)oracle_frame_count_s(8000, 16000.0f, fls[fi], 0.4375f, 0, &W, &L, &stp);
        float *y = malloc((W * 4097 + 1) * sizeof(float));
        (void)oracle_kaldi_spectrogram_s(y, x, 8000, 16000.0f, fls[fi], 0.4375f, 0, 0.0f, 1, 1, 0.97f, "povey", 0);
        (void)oracle_kaldi_mfcc_s(y, x, 8000, 16000.0f, fls[fi], 0.4375f, 0, 0.0f, 1, 1, 0.97f, "povey", 20.0f, 8000.0f, 23, 13, 22.0f, 0, 0);
        free(y);
    }
    /* the error returns leave everything alone */
    float y1[4];
    EXPECT(oracle_kaldi_mfcc_s(y1, x, 0, 16000.0f, 25.0f, 10.0f, 1, 0.0f, 1, 1, 0.97f, "povey", 20.0f, 8000.0f, 23, 13, 22.0f, 0, 0) == 1);
    EXPECT(oracle_kaldi_mfcc_s(y1, x, 4000, 16000.0f, 25.0f, 10.0f, 1, 0.0f, 1, 1, 0.97f, "povey", 20.0f, 8000.0f, 10, 13, 22.0f, 0, 0) == 1);
    EXPECT(oracle_kaldi_fbank_s(y1, x, 100, 16000.0f, 25.0f, 10.0f, 1, 0.0f, 1, 1, 0.97f, "povey", 0, 20.0f, 8000.0f, 23, 1, 0, 0) == 1);
    EXPECT(oracle_kaldi_mfcc_s(y1, x, 4000, 16000.0f, 25.0f, 10.0f, 1, 0.0f, 1, 1, 0.97f, "kaiser", 20.0f, 8000.0f, 23, 13, 22.0f, 0, 0) == 1);
    free(x); free(xd);
    printf("oracle sanitizer run: %d failure(s)\n", fails);
    return fails ? 1 : 0;
}
