// post_kernels.hip -- corpus post-processing on a batch's packed features, gfx950.
//
// The reference's only post-pass is mn0, the per-utterance mean subtraction (c/kaldi_fbank.c:355-370); these kernels
// generalise it to what Kaldi recipes run after feature extraction: cepstral mean / variance normalisation per utterance
// or with corpus-wide statistics (apply-cmvn), delta features (add-deltas) and frame splicing (splice-feats).  All are
// HBM-bound passes over frame-major W x D matrices packed back to back: consecutive lanes touch consecutive floats.
#include "speech_internal.hpp"

namespace speech {
namespace {

constexpr int NT = 256;                 // threads per workgroup

// Per-utterance column sums in double, one workgroup per utterance.  A block of Dc <= NT columns is covered by R = NT / Dc
// rows of threads (coalesced), each thread walks every R-th frame, the R partial sums of a column are added in row order:
// the result does not depend on scheduling.
template <typename T>
__global__ __launch_bounds__(NT) void utt_moments_kernel(const PostParams p)
{
    __shared__ double sh[2 * NT];
    const UttDesc utt = p.utts[blockIdx.x];
    const T *y = reinterpret_cast<const T *>(p.Yin) + utt.y_off;
    const long long W = (long long)utt.W;
    double *mom = p.mom + (size_t)blockIdx.x * 2 * p.D;
    const int tid = threadIdx.x;
    for (int c0 = 0; c0 < p.D; c0 += NT)
    {
        const int Dc = min(p.D - c0, NT), R = NT / Dc;
        const int r = tid / Dc, dc = tid - r * Dc;
        double s = 0.0, q = 0.0;
        if (r < R)
        {
            for (long long w = r; w < W; w += R) { const double v = (double)y[w * p.D + c0 + dc]; s += v; q += v * v; }
        }
        sh[tid] = s; sh[NT + tid] = q;
        __syncthreads();
        if (tid < Dc)
        {
            for (int k = 1; k < R; ++k) { s += sh[tid + k * Dc]; q += sh[NT + tid + k * Dc]; }
            mom[c0 + tid] = s; mom[p.D + c0 + tid] = q;
        }
        __syncthreads();
    }
}

// y := y * scale[d] + offset[d] (a multiply, then an add, each rounded in T: Kaldi's MulColsVec + AddVecToRows).
// mode 0: scale / offset given (corpus-wide statistics); mode 1: from the utterance's own moments -- mean removal, and
// division by the standard deviation if norm_vars (variance floored at 1e-20 as apply-cmvn does).
template <typename T>
__global__ __launch_bounds__(NT) void utt_affine_kernel(const PostParams p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char post_smem[];
    T *sc = reinterpret_cast<T *>(post_smem);
    T *of = sc + p.D;
    const UttDesc utt = p.utts[blockIdx.x];
    const long long W = (long long)utt.W;
    const long long row0 = (long long)blockIdx.y * p.rows_per_wg;
    if (row0 >= W) { return; }
    for (int d = threadIdx.x; d < p.D; d += NT)
    {
        double s = 1.0, o = 0.0;
        if (p.mode == 0) { s = p.scale[d]; o = p.offset[d]; }
        else
        {
            const double *mom = p.mom + (size_t)blockIdx.x * 2 * p.D;
            const double mean = mom[d] / (double)W;
            if (p.norm_vars)
            {
                double var = mom[p.D + d] / (double)W - mean * mean;
                if (var < 1.0e-20) { var = 1.0e-20; }
                s = 1.0 / sqrt(var);
            }
            o = -mean * s;
        }
        sc[d] = (T)s; of[d] = (T)o;
    }
    __syncthreads();
    const long long rows = min((long long)p.rows_per_wg, W - row0);
    T *y = reinterpret_cast<T *>(p.Y) + utt.y_off + (unsigned long long)row0 * p.D;
    const long long n = rows * p.D;
    int d = threadIdx.x % p.D;
    const int dstep = NT % p.D;
    for (long long i = threadIdx.x; i < n; i += NT)
    {
        T v = y[i] * sc[d];
        asm volatile("" : "+v"(v));                                          // keep the product rounded: no fma
        y[i] = v + of[d];
        d += dstep; if (d >= p.D) { d -= p.D; }
    }
}

// add-deltas: Z[t] = [ Y[t] | delta | delta-delta ... ], order k uses the k-fold convolution of the window
// [-w..w] / sum(j^2) with itself, evaluated on the input with the frame index clamped to the utterance
// (Kaldi feature-functions.cc, DeltaFeatures::Process).
template <typename T>
__global__ __launch_bounds__(NT) void add_deltas_kernel(const PostParams p)
{
    const UttDesc utt = p.utts[blockIdx.x];
    const long long W = (long long)utt.W;
    const long long row0 = (long long)blockIdx.y * p.rows_per_wg;
    if (row0 >= W) { return; }
    const long long rows = min((long long)p.rows_per_wg, W - row0);
    const T *y = reinterpret_cast<const T *>(p.Yin) + utt.y_off;
    const int K = p.order + 1, Dz = p.D * K;
    T *z = reinterpret_cast<T *>(p.Y) + utt.y_off * (unsigned long long)K;
    const long long n = rows * p.D;
    for (long long i = threadIdx.x; i < n; i += NT)
    {
        const long long t = row0 + i / p.D;
        const int d = (int)(i % p.D);
        for (int k = 0; k < K; ++k)
        {
            const int half = k * p.window;
            T acc = (T)0;
            for (int j = -half; j <= half; ++j)
            {
                long long tt = t + j;
                tt = tt < 0 ? 0 : (tt >= W ? W - 1 : tt);
                acc = fma((T)p.dscale[k][j + half], y[tt * p.D + d], acc);
            }
            z[t * Dz + k * p.D + d] = acc;
        }
    }
}

// splice-feats: Z[t] = [ Y[t-left] | ... | Y[t] | ... | Y[t+right] ], frame index clamped to the utterance
template <typename T>
__global__ __launch_bounds__(NT) void splice_kernel(const PostParams p)
{
    const UttDesc utt = p.utts[blockIdx.x];
    const long long W = (long long)utt.W;
    const long long row0 = (long long)blockIdx.y * p.rows_per_wg;
    if (row0 >= W) { return; }
    const long long rows = min((long long)p.rows_per_wg, W - row0);
    const T *y = reinterpret_cast<const T *>(p.Yin) + utt.y_off;
    const int K = p.left + p.right + 1, Dz = p.D * K;
    T *z = reinterpret_cast<T *>(p.Y) + utt.y_off * (unsigned long long)K + (unsigned long long)row0 * Dz;
    const long long n = rows * Dz;
    for (long long i = threadIdx.x; i < n; i += NT)
    {
        const long long t = row0 + i / Dz;
        const int c = (int)(i % Dz);
        long long tt = t + c / p.D - p.left;
        tt = tt < 0 ? 0 : (tt >= W ? W - 1 : tt);
        z[i] = y[tt * p.D + c % p.D];
    }
}

}  // namespace

hipError_t launch_post(int what, bool f64, const PostParams &p, uint32_t max_frames, hipStream_t s)
{
    if (p.nutt == 0) { return hipSuccess; }
    const uint32_t chunks = (max_frames + (uint32_t)p.rows_per_wg - 1u) / (uint32_t)p.rows_per_wg;
    const dim3 g1(p.nutt), g2(p.nutt, chunks ? chunks : 1u);
    const size_t lds = 2 * (size_t)p.D * (f64 ? sizeof(double) : sizeof(float));
    switch (what)
    {
        case POST_MOMENTS: if (f64) { hipLaunchKernelGGL(utt_moments_kernel<double>, g1, dim3(NT), 0, s, p); } else { hipLaunchKernelGGL(utt_moments_kernel<float>, g1, dim3(NT), 0, s, p); } break;
        case POST_AFFINE: if (f64) { hipLaunchKernelGGL(utt_affine_kernel<double>, g2, dim3(NT), lds, s, p); } else { hipLaunchKernelGGL(utt_affine_kernel<float>, g2, dim3(NT), lds, s, p); } break;
        case POST_DELTAS: if (f64) { hipLaunchKernelGGL(add_deltas_kernel<double>, g2, dim3(NT), 0, s, p); } else { hipLaunchKernelGGL(add_deltas_kernel<float>, g2, dim3(NT), 0, s, p); } break;
        case POST_SPLICE: if (f64) { hipLaunchKernelGGL(splice_kernel<double>, g2, dim3(NT), 0, s, p); } else { hipLaunchKernelGGL(splice_kernel<float>, g2, dim3(NT), 0, s, p); } break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}  // namespace speech
