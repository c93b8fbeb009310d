// synth_kernels.hip -- the synthetic corpus of SURVEY.md 8(d), generated on the device so that a 10k-hour benchmark corpus
// never exists on the host: an integer-only, counter-based generator (speech_internal.hpp, synth_sample), so the host
// (speech_amd/corpus.py, numpy) and the device produce the same samples bit for bit.  Bench / test support: the reference
// has no counterpart (it reads one tensor per call, src/kaldi_mfcc.cpp:283-289).
#include "speech_internal.hpp"

namespace speech {
namespace {

constexpr int NT = 256, CHUNK = 4096;   // a workgroup writes runs of 4096 samples of ONE utterance, lane after lane

template <typename T>
__global__ __launch_bounds__(NT) void synth_kernel(T *X, const UttDesc *utts, const unsigned long long *ids, unsigned long long seed)
{
    const unsigned u = blockIdx.x;
    const UttDesc utt = utts[u];
    const SynthUtt s = synth_utt(seed, ids ? ids[u] : (unsigned long long)u);
    T *x = X + utt.x_off;
    for (unsigned long long c0 = (unsigned long long)blockIdx.y * CHUNK; c0 < utt.n; c0 += (unsigned long long)gridDim.y * CHUNK)
    {
#pragma unroll 4
        for (int j = 0; j < CHUNK / NT; ++j)
        {
            const unsigned long long n = c0 + threadIdx.x + (unsigned)j * NT;
            if (n < utt.n) { x[n] = (T)synth_sample(s, (unsigned)n); }
        }
    }
}

}  // namespace

hipError_t launch_synth(void *X, int elem, const UttDesc *utts, const unsigned long long *ids, uint32_t nutt, unsigned long long max_n,
                        unsigned long long seed, hipStream_t s)
{
    if (nutt == 0) { return hipSuccess; }
    unsigned long long chunks = (max_n + CHUNK - 1) / CHUNK;
    if (chunks < 1) { chunks = 1; }
    if (chunks > 65535ull) { chunks = 65535ull; }                        // longer utterances: the workgroups stride over their chunks
    const dim3 grid(nutt, (unsigned)chunks);
    if (elem == 2) { hipLaunchKernelGGL(synth_kernel<short>, grid, dim3(NT), 0, s, (short *)X, utts, ids, seed); }
    else if (elem == 8) { hipLaunchKernelGGL(synth_kernel<double>, grid, dim3(NT), 0, s, (double *)X, utts, ids, seed); }
    else { hipLaunchKernelGGL(synth_kernel<float>, grid, dim3(NT), 0, s, (float *)X, utts, ids, seed); }
    return hipGetLastError();
}

}  // namespace speech
