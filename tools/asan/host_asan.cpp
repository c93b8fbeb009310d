// Host-side AddressSanitizer run of libspeech's host layer (validation, table building, mel-tile partition, batch geometry,
// plan cache, staging) -- device code is untouched.  Built by tools/asan/run.sh with -Xarch_host -fsanitize=address and run
// on the GPU box; exits non-zero on any ASan report or wrong return code.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <unistd.h>
#include <hip/hip_runtime.h>
#include "../../include/codee_speech.h"
#include "../../include/speech_hip.h"

static int fails = 0;
#define EXPECT(cond) do { if (!(cond)) { fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); ++fails; } } while (0)

int main()
{
    std::vector<float> x(48000);
    for (size_t i = 0; i < x.size(); ++i) { x[i] = 1000.0f * sinf(0.01f * (float)i) + (float)((i * 2654435761u) % 2001) - 1000.0f; }
    std::vector<double> xd(x.begin(), x.end());
    // drop-in calls over sample rates, mel-bin counts (every tile partition from 2 to 128 bins), cepstra, switches
    const float srs[] = {8000.0f, 16000.0f, 22050.0f, 44100.0f};
    for (float sr : srs)
    {
        const size_t N = 12000;
        for (size_t B = 2; B <= 140; B += (B < 24 ? 1 : 7))
        {
            const float hif = sr / 2;
            std::vector<float> y((N / 10 + 8) * (B + 2), 0.0f);
            EXPECT(codee::kaldi_fbank_s(y.data(), x.data(), N, sr, 25.0f, 10.0f, (int)(B & 1), 0.0f, 1, 1, 0.97f, "povey", 0, 20.0f, hif, B, 1, (int)((B >> 1) & 1), 0) == 0);
            const size_t C = B < 13 ? B : 13;
            EXPECT(codee::kaldi_mfcc_s(y.data(), x.data(), N, sr, 25.0f, 10.0f, 1, (B % 5 == 0) ? 1.0f : 0.0f, 1, 1, 0.97f, "hamming", 20.0f, hif, B, C, 22.0f, 0, (int)(B % 3 == 0)) == 0);
            if (B >= 13) { EXPECT(codee::kaldi_plp_s(y.data(), x.data(), N, sr, 25.0f, 10.0f, 0, 0.0f, 1, 0, 0.97f, "hann", 20.0f, hif, B, 1.0f / 3.0f, 13, 22.0f, 1.0f, 1, 0) == 0); }
        }
        std::vector<float> ys((12000 / 10 + 8) * 2049, 0.0f);
        EXPECT(codee::kaldi_spectrogram_s(ys.data(), x.data(), 12000, sr, 25.0f, 10.0f, 0, 0.0f, 0, 1, 0.0f, "rectangular", 0) == 0);
        std::vector<double> yd((12000 / 10 + 8) * 2049, 0.0);
        EXPECT(codee::kaldi_spectrogram_d(yd.data(), xd.data(), 12000, (double)sr, 25.0, 10.0, 1, 0.0, 1, 0, 0.97, "blackman", 1) == 0);
    }
    // frame geometry sweep on the tuned path: every zero-padding variant, odd lengths, shifts up to gaps between frames
    for (float fl = 16.1f; fl <= 32.0f; fl += 0.7f)
    {
        for (float fs = 2.0f; fs <= 40.0f; fs += 5.3f)
        {
            std::vector<float> y(48000 * 13, 0.0f);
            EXPECT(codee::kaldi_mfcc_s(y.data(), x.data(), 20000, 16000.0f, fl, fs, 0, 0.0f, 1, 1, 0.97f, "povey", 20.0f, 8000.0f, 23, 13, 22.0f, 0, 0) == 0);
            EXPECT(codee::kaldi_fbank_s(y.data(), x.data(), 9000, 8000.0f, fl, fs, 1, 0.5f, 1, 1, 0.97f, "povey", 1, 20.0f, 4000.0f, 23, 0, 1, 0) == 0);
        }
    }
    // 200 distinct option sets through the drop-in entry: the plan cache holds 16 and destroys what it evicts once the last
    // call using it has returned, so the device memory in use must not grow with the number of option sets
    {
        std::vector<float> y(48000 * 13, 0.0f);
        size_t f0 = 0, f1 = 0, tot = 0;
        EXPECT(codee::kaldi_mfcc_s(y.data(), x.data(), 16000, 16000.0f, 25.0f, 10.0f, 0, 0.0f, 1, 1, 0.97f, "povey", 20.0f, 8000.0f, 23, 13, 22.0f, 0, 0) == 0);
        EXPECT(hipMemGetInfo(&f0, &tot) == hipSuccess);
        long long grown = 0, grown200 = 0;
        for (int i = 0; i < 400; ++i)
        {
            EXPECT(codee::kaldi_mfcc_s(y.data(), x.data(), 16000, 16000.0f, 25.0f, 10.0f, 0, 0.0f, 1, 1, 0.97f, "povey", 20.0f + (float)i, 8000.0f - (float)(i % 7), 23 + (size_t)(i % 18), 13, 22.0f, 0, 0) == 0);
            if (i == 199 || i == 399)
            {
                EXPECT(hipMemGetInfo(&f1, &tot) == hipSuccess);
                grown = (long long)f0 - (long long)f1;
                if (i == 199) { grown200 = grown; }
                printf("device memory in use after %d option sets: %+lld bytes\n", i + 1, grown);
            }
        }
        // the second 200 sets add nothing: what is held is the cache of 16 plans.  Checked only with the sanitizer's quarantine
        // off (HOST_ASAN_MEMCHECK=1, second pass of run.sh): ROCm's ASan runtime quarantines freed DEVICE blocks too, which
        // reads as 199 KB of growth per option set until the quarantine (256 MB) is full
        if (getenv("HOST_ASAN_MEMCHECK")) { EXPECT(grown <= grown200 + (4ll << 20)); EXPECT(grown < (64ll << 20)); }
    }
    // error paths return 1 and leak nothing
    std::vector<float> y(4096, 0.0f);
    EXPECT(codee::kaldi_mfcc_s(y.data(), x.data(), 0, 16000.0f, 25.0f, 10.0f, 1, 0.0f, 1, 1, 0.97f, "povey", 20.0f, 8000.0f, 23, 13, 22.0f, 0, 0) == 1);
    EXPECT(codee::kaldi_mfcc_s(y.data(), x.data(), 4000, 16000.0f, 25.0f, 10.0f, 1, 0.0f, 1, 1, 0.97f, "nowindow", 20.0f, 8000.0f, 23, 13, 22.0f, 0, 0) == 1);
    EXPECT(codee::kaldi_mfcc_s(y.data(), x.data(), 4000, 16000.0f, 25.0f, 10.0f, 1, 0.0f, 1, 1, 0.97f, "povey", 20.0f, 8000.0f, 10, 13, 22.0f, 0, 0) == 1);
    EXPECT(codee::kaldi_fbank_s(y.data(), x.data(), 100, 16000.0f, 25.0f, 10.0f, 1, 0.0f, 1, 1, 0.97f, "povey", 0, 20.0f, 8000.0f, 23, 1, 0, 0) == 1);      // snip_edges && L > N
    // the batch ABI with ragged lengths, post-processing included
    speech_opts o; speech_opts_default(&o, SPEECH_MFCC);
    o.snip_edges = 0; o.dc0 = 1; o.raw_energy = 1;
    speech_plan *plan = nullptr;
    EXPECT(speech_plan_create(&plan, &o, 0) == 0);
    if (plan)
    {
        const size_t lens[] = {16000, 1, 399, 400, 401, 47999, 160};
        speech_batch *b = nullptr;
        EXPECT(speech_batch_create(&b, plan, 7, lens, nullptr, nullptr) == 0);
        if (b)
        {
            size_t ns = 0, no = 0, nf = 0;
            EXPECT(speech_batch_totals(b, &ns, &no, &nf) == 0);
            std::vector<double> sum(13), sq(13); unsigned long long frames = 0;
            EXPECT(speech_batch_stats(b, nullptr, sum.data(), sq.data(), &frames, nullptr) == 1);      // null device pointer
            speech_batch_destroy(b);
        }
        // a batch of 200 utterances on the device: run, statistics through the two-level tree, CMVN, deltas, splice + subsample
        std::vector<size_t> many(200);
        size_t tot = 0;
        for (size_t i = 0; i < many.size(); ++i) { many[i] = 160u * (1u + (7u * i) % 5u); tot += many[i]; }
        EXPECT(speech_batch_create(&b, plan, many.size(), many.data(), nullptr, nullptr) == 0);
        if (b)
        {
            size_t ns = 0, no = 0, nf = 0, zr = 0;
            EXPECT(speech_batch_totals(b, &ns, &no, &nf) == 0 && ns == tot);
            std::vector<size_t> rows(many.size());
            EXPECT(speech_batch_subsample_rows(b, 3, 1, rows.data(), &zr) == 0);
            EXPECT(speech_batch_subsample_rows(b, 3, 3, nullptr, &zr) == 1);
            float *dX = nullptr, *dY = nullptr, *dZ = nullptr;
            EXPECT(hipMalloc((void **)&dX, ns * 4) == hipSuccess && hipMalloc((void **)&dY, no * 4) == hipSuccess && hipMalloc((void **)&dZ, no * 3 * 4) == hipSuccess);
            if (dX && dY && dZ)
            {
                std::vector<float> xs(ns);
                for (size_t i = 0; i < ns; ++i) { xs[i] = x[i % x.size()]; }
                EXPECT(hipMemcpy(dX, xs.data(), ns * 4, hipMemcpyHostToDevice) == hipSuccess);
                EXPECT(speech_batch_run(b, dX, dY, nullptr) == 0);
                std::vector<double> sum(13), sq(13); unsigned long long frames = 0;
                EXPECT(speech_batch_stats(b, dY, sum.data(), sq.data(), &frames, nullptr) == 0 && frames == nf);
                EXPECT(speech_batch_cmvn(b, dY, 1, nullptr) == 0);
                EXPECT(speech_batch_affine(b, dY, sum.data(), sq.data(), nullptr) == 0);
                EXPECT(speech_batch_add_deltas(b, dY, dZ, 2, 2, nullptr) == 0);
                EXPECT(speech_batch_splice(b, dY, dZ, 1, 1, nullptr) == 0);
                EXPECT(speech_batch_splice_subsample(b, dY, dZ, 3, 3, 3, 1, nullptr) == 0);
                EXPECT(speech_batch_splice_subsample(b, dY, dZ, 0, 0, 2, 0, nullptr) == 0);       // the offset table is re-made
                EXPECT(speech_batch_splice_subsample(b, dY, dZ, 0, 0, 2, 2, nullptr) == 1);
                EXPECT(hipDeviceSynchronize() == hipSuccess);
            }
            (void)hipFree(dX); (void)hipFree(dY); (void)hipFree(dZ);
            speech_batch_destroy(b);
        }
        // ragged list from host memory through the two-stream pipeline, one utterance per chunk and all in one
        {
            std::vector<const void *> xp; std::vector<void *> yp; std::vector<size_t> nn;
            std::vector<std::vector<float>> outs;
            for (size_t i = 0; i < 9; ++i) { nn.push_back(400 + 1777 * i); outs.emplace_back(((nn.back() + 80) / 160 + 1) * 13, 0.0f); }
            for (size_t i = 0; i < 9; ++i) { xp.push_back(x.data() + 100 * i); yp.push_back(outs[i].data()); }
            EXPECT(speech_plan_run_host(plan, 9, xp.data(), nn.data(), yp.data(), 0, 1, nullptr) == 0);
            EXPECT(speech_plan_run_host(plan, 9, xp.data(), nn.data(), yp.data(), 0, 0, nullptr) == 0);
            nn[4] = 0;
            EXPECT(speech_plan_run_host(plan, 9, xp.data(), nn.data(), yp.data(), 0, 30000, nullptr) == 1);       // an invalid utterance in a later chunk
            nn[4] = 5000;
            EXPECT(speech_plan_run_host(plan, 9, xp.data(), nn.data(), yp.data(), 0, 30000, nullptr) == 0);
            speech_host_release();
        }
        // 16-bit PCM from host memory
        std::vector<short> x16(20000);
        for (size_t i = 0; i < x16.size(); ++i) { x16[i] = (short)x[i]; }
        std::vector<float> y16(200 * 13, 0.0f);
        EXPECT(speech_extract_pcm16(&o, x16.data(), x16.size(), y16.data()) == 0);
        EXPECT(speech_extract_pcm16(&o, nullptr, x16.size(), y16.data()) == 1);
        EXPECT(speech_extract_pcm16(nullptr, x16.data(), x16.size(), y16.data()) == 1);
        speech_plan_destroy(plan);
    }
    printf("host asan run: %d failure(s)\n", fails);
    fflush(stdout);
    // leave without the exit-time finalizers: with ROCm 7.2 the sanitizer runtime's device-allocator hook aborts inside
    // libamdhip64's own teardown (CHECK dev_runtime_unloaded_), after main has returned -- not this library's code
    _exit(fails ? 1 : 0);
}
