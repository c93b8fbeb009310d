// post_kernels.hip -- corpus post-processing on a batch's packed features, gfx950.
//
// The reference's only post-pass is mn0, the per-utterance mean subtraction (c/kaldi_fbank.c:355-370); these kernels
// generalise it to what Kaldi recipes run after feature extraction: cepstral mean / variance normalisation per utterance
// or with corpus-wide statistics (apply-cmvn), delta features (add-deltas), frame splicing (splice-feats) and frame
// subsampling (subsample-feats, fused with the splice).  All are
// HBM-bound passes over frame-major W x D matrices packed back to back: consecutive lanes touch consecutive floats.
#include "speech_internal.hpp"
#include <type_traits>

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

// Sums of groups of POST_REDUCE_GROUP consecutive rows of a [n][cols] array of doubles, each group added in row order:
// the corpus-wide statistics as a tree of fixed shape over the utterances, the same sums on every run.
__global__ __launch_bounds__(NT) void reduce_rows_kernel(const double *__restrict__ in, double *__restrict__ out, unsigned long long n, int cols)
{
    const unsigned long long u0 = (unsigned long long)blockIdx.x * POST_REDUCE_GROUP;
    const int cnt = (int)min((unsigned long long)POST_REDUCE_GROUP, n - u0);
    for (int c = threadIdx.x; c < cols; c += NT)
    {
        double s = 0.0;
        for (int k = 0; k < cnt; ++k) { s += in[(u0 + k) * cols + c]; }
        out[(unsigned long long)blockIdx.x * cols + c] = s;
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
// (Kaldi feature-functions.cc, DeltaFeatures::Process).  A workgroup takes rows_per_wg frames; a thread walks the chunk's
// elements (frame r, column d, kept incrementally: no division per element), reads the 2 order window + 1 neighbouring
// frames once (consecutive lanes on consecutive floats; the re-reads hit the caches) and feeds every order's filter from
// them, each order summing its taps in ascending order.
// ORDER, WINDOW >= 0: fixed at compile time (Kaldi's default 2, 2): the neighbouring frames' loads are issued together and
// the filter loops unroll; ORDER < 0: any order / window from the arguments, one frame at a time.
template <typename T, int ORDER, int WINDOW>
__global__ __launch_bounds__(NT) void add_deltas_kernel(const PostParams p)
{
    __shared__ T taps[POST_MAX_ORDER + 1][POST_MAX_TAPS + 1];              // the filters in T: the kernel arguments hold them in double
    const UttDesc utt = p.utts[blockIdx.x];
    const long long W = (long long)utt.W;
    const long long row0 = (long long)blockIdx.y * p.rows_per_wg;
    if (row0 >= W) { return; }
    for (int i = threadIdx.x; i < (POST_MAX_ORDER + 1) * POST_MAX_TAPS; i += NT) { taps[i / POST_MAX_TAPS][i % POST_MAX_TAPS] = (T)p.dscale[i / POST_MAX_TAPS][i % POST_MAX_TAPS]; }
    __syncthreads();
    constexpr bool FIXED = ORDER >= 0;
    const int order = FIXED ? ORDER : p.order, window = FIXED ? WINDOW : p.window;
    const int rows = (int)min((long long)p.rows_per_wg, W - row0);
    const int D = p.D, K = order + 1, Dz = D * K, maxhalf = order * window;
    const T *__restrict__ y0 = reinterpret_cast<const T *>(p.Yin) + utt.y_off + (unsigned long long)row0 * D;
    T *__restrict__ z0 = reinterpret_cast<T *>(p.Y) + utt.y_off * (unsigned long long)K + (unsigned long long)row0 * Dz;
    // frame index relative to row0, clamped to the utterance (the bounds only matter within maxhalf of the chunk)
    const int lo = (int)max(-row0, (long long)-64), hi = (int)min(W - 1 - row0, (long long)rows + 64);
    const int n = rows * D, rstep = NT / D, dstep = NT % D;
    int r = (int)threadIdx.x / D, d = (int)threadIdx.x - r * D;
    for (int i = threadIdx.x; i < n; i += NT)
    {
        T acc[POST_MAX_ORDER + 1];
#pragma unroll
        for (int k = 0; k <= POST_MAX_ORDER; ++k) { acc[k] = (T)0; }
        if (FIXED)
        {
            constexpr int MH = FIXED ? ORDER * WINDOW : 0;
            T v[2 * MH + 1];
#pragma unroll
            for (int j = -MH; j <= MH; ++j) { v[j + MH] = y0[min(max(r + j, lo), hi) * D + d]; }
#pragma unroll
            for (int j = -MH; j <= MH; ++j)
            {
#pragma unroll
                for (int k = 0; k <= (FIXED ? ORDER : 0); ++k)
                {
                    if ((j < 0 ? -j : j) <= k * WINDOW) { acc[k] = fma(taps[k][j + k * WINDOW], v[j + MH], acc[k]); }
                }
            }
        }
        else
        {
            for (int j = -maxhalf; j <= maxhalf; ++j)
            {
                const T v = y0[min(max(r + j, lo), hi) * D + d];
                const int aj = j < 0 ? -j : j;
#pragma unroll
                for (int k = 0; k <= POST_MAX_ORDER; ++k)
                {
                    if (k < K && aj <= k * window) { acc[k] = fma(taps[k][j + k * window], v, acc[k]); }
                }
            }
        }
        T *zr = z0 + (size_t)r * Dz + d;
#pragma unroll
        for (int k = 0; k <= POST_MAX_ORDER; ++k) { if (k < K) { zr[k * D] = acc[k]; } }
        r += rstep; d += dstep;
        if (d >= D) { d -= D; ++r; }
    }
}

// splice-feats: Z[t] = [ Y[t-left] | ... | Y[t] | ... | Y[t+right] ], frame index clamped to the utterance; SUB: followed by
// subsample-feats in the same pass (output row j of an utterance is the spliced frame sub_off + j sub_n; the utterances'
// outputs packed back to back, zrow[u] = rows of the utterances before u).
// Away from the utterance's ends a spliced row is one contiguous run of Y: (left + right + 1) D elements from frame
// t - left on, so the pass is a copy with overlapping sources.  E is the unit of that copy: T, or 16 bytes of T when D and
// the utterance offsets allow it.  One wave per output row (several rows per wave when a row is shorter than 64 units).
template <typename E, bool SUB>
__device__ __forceinline__ void splice_rows(const PostParams &p, int D, unsigned long long y_off, unsigned long long z_off)
{
    const UttDesc utt = p.utts[blockIdx.x];
    const long long W = (long long)utt.W;
    const long long Wz = !SUB ? W : (W > p.sub_off ? (W - p.sub_off + p.sub_n - 1) / p.sub_n : 0);
    const long long row0 = (long long)blockIdx.y * p.rows_per_wg;
    if (row0 >= Wz) { return; }
    const int rows = (int)min((long long)p.rows_per_wg, Wz - row0);
    const int K = p.left + p.right + 1, Dz = D * K;
    const E *y = reinterpret_cast<const E *>(p.Yin) + y_off;
    E *z = reinterpret_cast<E *>(p.Y) + z_off + (unsigned long long)row0 * Dz;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int rpw = Dz >= 64 ? 1 : 64 / Dz;                                  // rows per wave and pass
    const int lr = Dz >= 64 ? 0 : lane / Dz, lc = lane - lr * Dz;
    if (lr >= rpw) { return; }
    for (int r = wave * rpw + lr; r < rows; r += (NT / 64) * rpw)
    {
        const long long t = SUB ? (long long)p.sub_off + (row0 + r) * p.sub_n : row0 + r;
        E *zr = z + (size_t)r * Dz;
        if (t - p.left >= 0 && t + p.right < W)
        {
            const E *src = y + (t - p.left) * D;
            for (int c = lc; c < Dz; c += 64) { zr[c] = src[c]; }
        }
        else
        {
            for (int c = lc; c < Dz; c += 64)
            {
                const int k = c / D;
                long long tt = t + k - p.left;
                tt = tt < 0 ? 0 : (tt >= W ? W - 1 : tt);
                zr[c] = y[tt * D + (c - k * D)];
            }
        }
    }
}

template <typename T, bool SUB, bool VEC>
__global__ __launch_bounds__(NT) void splice_kernel(const PostParams p)
{
    const UttDesc utt = p.utts[blockIdx.x];
    const unsigned long long K = (unsigned long long)(p.left + p.right + 1);
    constexpr int VW = 16 / (int)sizeof(T);
    typedef typename std::conditional<sizeof(T) == 4, float4, double2>::type V;
    if (VEC)
    {
        const int Dv = p.D / VW;
        splice_rows<V, SUB>(p, Dv, utt.y_off / VW, SUB ? p.zrow[blockIdx.x] * K * Dv : utt.y_off / VW * K);
    }
    else { splice_rows<T, SUB>(p, p.D, utt.y_off, SUB ? p.zrow[blockIdx.x] * K * p.D : utt.y_off * K); }
}

template <bool SUB>
void launch_splice(bool f64, const PostParams &p, dim3 g, hipStream_t s)
{
    if (f64)
    {
        if (p.vec16) { hipLaunchKernelGGL((splice_kernel<double, SUB, true>), g, dim3(NT), 0, s, p); }
        else { hipLaunchKernelGGL((splice_kernel<double, SUB, false>), g, dim3(NT), 0, s, p); }
    }
    else
    {
        if (p.vec16) { hipLaunchKernelGGL((splice_kernel<float, SUB, true>), g, dim3(NT), 0, s, p); }
        else { hipLaunchKernelGGL((splice_kernel<float, SUB, false>), g, dim3(NT), 0, s, p); }
    }
}

}  // namespace

hipError_t launch_reduce_rows(const double *in, double *out, size_t n, int cols, hipStream_t s)
{
    if (n == 0) { return hipSuccess; }
    hipLaunchKernelGGL(reduce_rows_kernel, dim3((unsigned)((n + POST_REDUCE_GROUP - 1) / POST_REDUCE_GROUP)), dim3(NT), 0, s, in, out, (unsigned long long)n, cols);
    return hipGetLastError();
}

hipError_t post_prepare(bool f64, int D)
{
    const size_t lds = (size_t)2 * (size_t)D * (f64 ? 8u : 4u);
    if (lds <= 48u * 1024u) { return hipSuccess; }
    const void *k = f64 ? reinterpret_cast<const void *>(&utt_affine_kernel<double>) : reinterpret_cast<const void *>(&utt_affine_kernel<float>);
    // once per (device, kernel), to the widest feature row the host layer admits (the spectrogram of an 8192-point frame:
    // 4097 columns; fbank with energy: 4097): the attribute is the kernel's, not the plan's
    return opt_in_lds_once(k, (int)(2u * ((size_t)GENERIC_MAX_NFFT / 2u + 1u) * 8u));
}

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
        case POST_DELTAS:
            if (p.order == 2 && p.window == 2)
            {
                if (f64) { hipLaunchKernelGGL((add_deltas_kernel<double, 2, 2>), g2, dim3(NT), 0, s, p); } else { hipLaunchKernelGGL((add_deltas_kernel<float, 2, 2>), g2, dim3(NT), 0, s, p); }
            }
            else if (f64) { hipLaunchKernelGGL((add_deltas_kernel<double, -1, -1>), g2, dim3(NT), 0, s, p); }
            else { hipLaunchKernelGGL((add_deltas_kernel<float, -1, -1>), g2, dim3(NT), 0, s, p); }
            break;
        case POST_SPLICE: launch_splice<false>(f64, p, g2, s); break;
        case POST_SPLICE_SUB: launch_splice<true>(f64, p, g2, s); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}  // namespace speech
