// generic_kernels.hip -- any-size, any-precision per-frame pipeline (gfx950).
//
// One 256-thread workgroup walks a tile of consecutive frames of one utterance; the frame lives in
// LDS from the gather to the feature epilogue.  This path serves the *_d entry points, frame lengths
// without a tuned kernel (nfft other than 256, 512, 1024) and odd frame shifts; the tuned fp32 kernels are in
// fast_spectrogram.hip / fast_mel.hip.  Pipeline order and arithmetic follow the reference's per-frame loop
// (c/kaldi_mfcc.c:213-334; spectrogram epilogue c/kaldi_spectrogram.c:236-251; fbank epilogue
// c/kaldi_fbank.c:290-350; PLP tail per SURVEY.md 8(a) row P).
#include "speech_internal.hpp"
#include <cfloat>

namespace speech {

namespace {

template <typename T> struct M;
template <> struct M<float> {
    // natural log of a positive normal float (every caller floors its argument at FLT_EPSILON): exponent and mantissa split,
    // so that the hardware log2 (1 ulp of its RESULT) only sees [0.5, 1) and the one rounding left is the result's own.
    // The library logf measured ~1 ulp at |log| ~ 20, i.e. 2e-6 on a log-power value: twice the rms noise of the whole
    // float32 pipeline on the CPU (tools/noise_check.py); this form is the tuned kernels' (fast_common.hpp, pos_log).
    static __device__ __forceinline__ float log(float x)
    {
        const float m = __builtin_amdgcn_frexp_mantf(x);
        const float e = (float)__builtin_amdgcn_frexp_expf(x);
        return fmaf(e, 0.69314718055994530942f, __log2f(m) * 0.69314718055994530942f);
    }
    static __device__ __forceinline__ float sqrt(float x) { return sqrtf(x); }
    static __device__ __forceinline__ float pow(float x, float y) { return powf(x, y); }
    static __device__ __forceinline__ float cos(float x) { return cosf(x); }
    static __device__ __forceinline__ float sin(float x) { return sinf(x); }
};
template <> struct M<double> {
    static __device__ __forceinline__ double log(double x) { return ::log(x); }
    static __device__ __forceinline__ double sqrt(double x) { return ::sqrt(x); }
    static __device__ __forceinline__ double pow(double x, double y) { return ::pow(x, y); }
    static __device__ __forceinline__ double cos(double x) { return ::cos(x); }
    static __device__ __forceinline__ double sin(double x) { return ::sin(x); }
};

constexpr int NT = 256;

// Sum over the workgroup; every thread returns the same value.  `red` holds NT elements.
template <typename T, int NTH>
__device__ __forceinline__ T block_sum(T v, T *red)
{
    if (NTH == 64)
    {
        // one wave: the same pairing as the LDS tree below (lane i takes i + 32, then i + 16, ...), so the same bits,
        // without LDS or barriers
#pragma unroll
        for (int s = 32; s > 0; s >>= 1) { v += __shfl_xor(v, s, 64); }
        return v;
    }
    const int tid = threadIdx.x;
    __syncthreads();
    red[tid] = v;
    __syncthreads();
    for (int s = NTH / 2; s > 0; s >>= 1)
    {
        if (tid < s) { red[tid] += red[tid + s]; }
        __syncthreads();
    }
    return red[0];
}

template <typename T>
__device__ __forceinline__ T floor_log(T v)
{
    const T eps = (T)FLT_EPSILON;
    return (v < eps) ? M<T>::log(eps) : M<T>::log(v);
}

// NTH threads per workgroup: 256 for long transforms; 64 (one wave: its barriers cost next to nothing, and four times as
// many frames are in flight per CU) for nfft <= 1024 -- 1.4-1.9x faster there (tools/generic_time.py).
template <typename T, int NTH>
__global__ __launch_bounds__(NTH) void generic_frames_kernel(const FrameParams<T> p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int tid = threadIdx.x;
    const int L = p.L, nfft = p.nfft, F = p.F, B = p.B, C = p.C, P = p.P;
    T *re = reinterpret_cast<T *>(smem_raw);
    T *im = re + nfft;
    T *red = im + nfft;            // NT
    T *mel = red + NT;             // B (red is sized for the larger workgroup)
    T *dup = mel + B;              // B+2
    T *ac = dup + (B + 2);         // P+1
    T *lpc = ac + (P + 1);         // P+1
    T *tmp = lpc + (P + 1);        // P+1
    T *cep = tmp + (P + 1);        // P+1
    T *sc0 = cep + (P + 1);        // 1
    // one-wave workgroups (nfft <= 1024) keep the twiddles and the window in LDS: the radix-2 stages read a twiddle per
    // butterfly, and from global memory each stage of each frame pays an L1/L2 latency for it
    const T *twr = p.tw_re, *twi = p.tw_im, *wn = p.win;
    if (NTH == 64 && sizeof(T) == 4)                                        // f64: the extra LDS costs more in occupancy than it saves
    {
        T *c_twr = sc0 + 4, *c_twi = c_twr + nfft / 2, *c_win = c_twi + nfft / 2;
        for (int i = tid; i < nfft / 2; i += NTH) { c_twr[i] = p.tw_re[i]; c_twi[i] = p.tw_im[i]; }
        for (int i = tid; i < L; i += NTH) { c_win[i] = p.win[i]; }
        twr = c_twr; twi = c_twi; wn = c_win;
    }

    const TileDesc tile = p.tiles[blockIdx.x];
    const UttDesc utt = p.utts[tile.utt];
    const T *x = p.X + utt.x_off;
    const short *x16 = p.X16 ? p.X16 + utt.x_off : nullptr;
    const long long N = (long long)utt.n;
    const T eps = (T)FLT_EPSILON;

    for (uint32_t fi = 0; fi < tile.nframes; ++fi)
    {
        const long long w = (long long)tile.frame0 + fi;
        T *y = p.Y + utt.y_off + (unsigned long long)w * (unsigned long long)p.D;
        // frame start: c/kaldi_mfcc.c:81-82,216-240
        const long long start = p.snip_edges ? w * p.stp : (long long)(p.stp / 2) - (long long)(L / 2) + w * p.stp;
        __syncthreads();
        if (start >= 0 && start + L <= N)                                   // (uniform) an interior frame: no reflection, 32-bit indexing
        {
            const T *xf = x + start;
            const short *xf16 = x16 ? x16 + start : nullptr;
            for (int l = tid; l < nfft; l += NTH) { re[l] = (l < L) ? (xf16 ? (T)xf16[l] : xf[l]) : (T)0; }
        }
        else
        for (int l = tid; l < nfft; l += NTH)
        {
            T v = (T)0;
            if (l < L)
            {
                long long n = start + l;
                while (n < 0 || n >= N) { n = (n < 0) ? -n - 1 : 2 * N - 1 - n; }   // reflect: c/kaldi_mfcc.c:228
                v = x16 ? (T)x16[n] : x[n];
            }
            re[l] = v;
        }
        __syncthreads();
        if (p.do_dither)
        {
            // Gaussian pairs by Box-Muller as c/kaldi_mfcc.c:243-262, from a counter-based hash (speech_internal.hpp)
            const unsigned long long key = dither_frame_key(mix64(p.seed ^ ((unsigned long long)tile.uid << 32)), (unsigned long long)w);
            for (int i = tid; 2 * i < L; i += NTH)
            {
                T g0, g1;
                dither_pair(dither_bits(key, (unsigned)i), p.dither, g0, g1);
                re[2 * i] += g0;
                if (2 * i + 1 < L) { re[2 * i + 1] += g1; }
            }
            __syncthreads();
        }

        T rawe = (T)0;
        if (p.want_pre && p.pre_before_dc)                 // kaldi_mfcc_d only: c/kaldi_mfcc.c:576-582
        {
            T s = (T)0;
            for (int l = tid; l < L; l += NTH) { s += re[l] * re[l]; }
            rawe = block_sum<T, NTH>(s, red);
            if (rawe < eps) { rawe = eps; }
        }
        if (p.dc0)                                         // c/kaldi_mfcc.c:265-271
        {
            T s = (T)0;
            for (int l = tid; l < L; l += NTH) { s += re[l]; }
            const T mn = block_sum<T, NTH>(s, red) / (T)L;
            for (int l = tid; l < L; l += NTH) { re[l] -= mn; }
            __syncthreads();
        }
        if (p.want_pre && !p.pre_before_dc)                // c/kaldi_mfcc.c:274-279
        {
            T s = (T)0;
            for (int l = tid; l < L; l += NTH) { s += re[l] * re[l]; }
            rawe = block_sum<T, NTH>(s, red);
            if (rawe < eps) { rawe = eps; }
        }
        // pre-emphasis (closed form of the backwards in-place loop, c/kaldi_mfcc.c:282-288) and window (:291)
        for (int l = tid; l < nfft; l += NTH)
        {
            T v = (T)0;
            if (l < L)
            {
                v = re[l];
                if (p.do_preemph) { v -= p.preemph * ((l > 0) ? re[l - 1] : re[0]); }
                v *= wn[l];
            }
            im[l] = v;
        }
        __syncthreads();
        if (p.want_post)                                   // c/kaldi_mfcc.c:295-301
        {
            T s = (T)0;
            for (int l = tid; l < L; l += NTH) { s += im[l] * im[l]; }
            rawe = block_sum<T, NTH>(s, red);
            if (rawe < eps) { rawe = eps; }
        }
        // FFT (replaces the FFTW R2HC call, c/kaldi_mfcc.c:304) of the zero-padded real frame as a complex FFT of half the
        // size: z[m] = x[2m] + i x[2m+1], radix-2 DIT on n2 = nfft / 2 points in (zr, zi) = the two halves of `re`
        const int n2 = nfft >> 1, lg2 = p.log2n - 1;
        T *zr = re, *zi = re + n2;
        T *pw = im;                                        // the power spectrum replaces the windowed frame
        if (nfft == 1)
        {
            if (tid == 0) { const T v = im[0]; pw[0] = v * v; }
            __syncthreads();
        }
        else
        {
        for (int m = tid; m < n2; m += NTH)
        {
            const int r = (lg2 > 0) ? (int)(__brev((unsigned)m) >> (32 - lg2)) : 0;
            zr[r] = im[2 * m]; zi[r] = im[2 * m + 1];
        }
        __syncthreads();
        // two radix-2 stages per pass over LDS: the four points a, b = a + half, c = a + 2 half, d = a + 3 half go through
        // stage `half` (pairs a-b and c-d) and stage `2 half` (pairs a-c and b-d) in registers: half the LDS traffic and
        // barriers of one stage per pass.  The twiddle of span `half` is exp(-2 pi i k / (2 half)) = tw[k nfft / (2 half)].
        int half = 1;
        if (lg2 & 1)
        {
            for (int i = tid; i < n2 / 2; i += NTH)                        // lone first stage: twiddle 1
            {
                const int a = i << 1, b = a + 1;
                const T ar = zr[a], ai = zi[a], br = zr[b], bi = zi[b];
                zr[b] = ar - br; zi[b] = ai - bi;
                zr[a] = ar + br; zi[a] = ai + bi;
            }
            __syncthreads();
            half = 2;
        }
        int lh = (lg2 & 1) ? 1 : 0;                                         // log2(half)
        for (; half < n2; half <<= 2, lh += 2)
        {
            // twiddle strides nfft / (2 half) and nfft / (4 half) are powers of two: shifts, not 32-bit multiplies (quarter rate)
            const int sh1 = p.log2n - lh - 1, sh2 = sh1 - 1;
            for (int i = tid; i < n2 / 4; i += NTH)
            {
                const int k = i & (half - 1);
                const int a = ((i - k) << 2) + k, b = a + half, c = b + half, d = c + half;
                const T c1 = twr[k << sh1], s1 = twi[k << sh1];
                const T c2 = twr[k << sh2], s2 = twi[k << sh2];
                const T c3 = twr[(k + half) << sh2], s3 = twi[(k + half) << sh2];
                T ar = zr[a], ai = zi[a], br = zr[b], bi = zi[b], cr = zr[c], ci = zi[c], dr = zr[d], di = zi[d];
                // stage `half`
                T tr = br * c1 - bi * s1, ti = br * s1 + bi * c1;
                br = ar - tr; bi = ai - ti; ar = ar + tr; ai = ai + ti;
                tr = dr * c1 - di * s1; ti = dr * s1 + di * c1;
                dr = cr - tr; di = ci - ti; cr = cr + tr; ci = ci + ti;
                // stage `2 half`
                tr = cr * c2 - ci * s2; ti = cr * s2 + ci * c2;
                cr = ar - tr; ci = ai - ti; ar = ar + tr; ai = ai + ti;
                tr = dr * c3 - di * s3; ti = dr * s3 + di * c3;
                dr = br - tr; di = bi - ti; br = br + tr; bi = bi + ti;
                zr[a] = ar; zi[a] = ai; zr[b] = br; zi[b] = bi; zr[c] = cr; zi[c] = ci; zr[d] = dr; zi[d] = di;
            }
            __syncthreads();
        }
        // untangle + power, c/kaldi_mfcc.c:307-310: with A = Z[k] + conj Z[n2-k], B = Z[k] - conj Z[n2-k] and
        // w = exp(-2 pi i k / nfft), X[k] = (A - i w B) / 2.  X[0] = Zr[0] + Zi[0] and X[nfft/2] = Zr[0] - Zi[0] are real
        // (the reference never adds their imaginary parts).
        for (int k = tid; k < F; k += NTH)
        {
            T v;
            if (k == 0) { const T x0 = zr[0] + zi[0]; v = x0 * x0; }
            else if (k == n2) { const T xn = zr[0] - zi[0]; v = xn * xn; }
            else
            {
                const T pr = zr[k], pi = zi[k], qr = zr[n2 - k], qi = zi[n2 - k];
                const T Ar = pr + qr, Ai = pi - qi, Br = pr - qr, Bi = pi + qi;
                const T c = twr[k], s = twi[k];
                const T xr = Ar + (c * Bi + s * Br), xi = Ai - (c * Br - s * Bi);
                v = (T)0.25 * (xr * xr + xi * xi);
            }
            pw[k] = v;
        }
        __syncthreads();
        }

        if (p.kind == SPEECH_SPECTROGRAM)
        {
            // c/kaldi_spectrogram.c:236-251
            for (int k = tid; k < F; k += NTH)
            {
                const T v = (k == 0) ? rawe : ((k == F - 1) ? pw[F - 2] : pw[k]);
                y[k] = floor_log(v);
            }
            continue;
        }

        // sparse mel sums, c/kaldi_mfcc.c:313-320 / c/kaldi_fbank.c:312-345
        for (int b = tid; b < B; b += NTH)
        {
            const T *wgt = p.f2b + p.mel_off[b];
            const int beg = p.mel_beg[b], num = p.mel_num[b];
            T sm = (T)0;
            for (int f = 0; f < num; ++f)
            {
                T pv = pw[beg + f];
                if (p.amp) { pv = M<T>::sqrt(pv); }
                sm += pv * wgt[f];
            }
            mel[b] = sm;
        }
        __syncthreads();

        if (p.kind == SPEECH_FBANK)
        {
            const int eo = p.use_energy ? 1 : 0;
            for (int b = tid; b < B; b += NTH)
            {
                const int src = (b == B - 1 && B >= 2) ? B - 2 : b;        // c/kaldi_fbank.c:350
                const T v = mel[src];
                T out = p.lg ? floor_log(v) : v;
                if (B == 1 && eo) { out = M<T>::log(rawe); }
                y[eo + b] = out;
            }
            if (eo && tid == 0) { y[0] = M<T>::log(rawe); }                  // c/kaldi_fbank.c:299
            continue;
        }

        if (p.kind == SPEECH_MFCC)
        {
            for (int b = tid; b < B; b += NTH) { mel[b] = floor_log(mel[b]); }   // c/kaldi_mfcc.c:318
            __syncthreads();
            for (int c = tid; c < C; c += NTH)
            {
                const int cc = (c == C - 1 && C >= 2) ? C - 2 : c;         // c/kaldi_mfcc.c:333
                const T *d = p.dct + (size_t)cc * B;
                T acc = (T)0;
                for (int b = 0; b < B; ++b) { acc += mel[b] * d[b]; }
                T out = p.lift[cc] * acc;                                  // c/kaldi_mfcc.c:327-329
                if (p.use_energy && cc == 0) { out = M<T>::log(rawe); }
                y[c] = out;
            }
            continue;
        }

        // PLP tail
        for (int j = tid; j < B + 2; j += NTH)
        {
            const int b = (j == 0) ? 0 : ((j == B + 1) ? B - 1 : j - 1);
            const T v = mel[b] * p.eql[b];
            T cv;
            if (p.compress_mode == 1) { cv = v; }
            else if (p.compress_mode == 2) { cv = M<T>::sqrt(v); }
            else { cv = M<T>::pow(v, p.compress); }
            dup[j] = cv;
        }
        __syncthreads();
        for (int i = tid; i <= P; i += NTH)
        {
            const T *row = p.idft + (size_t)i * (B + 2);
            T acc = (T)0;
            for (int j = 0; j < B + 2; ++j) { acc += row[j] * dup[j]; }
            ac[i] = acc;
        }
        __syncthreads();
        if (tid == 0)
        {
            T E = ac[0];
            for (int i = 0; i < P; ++i)
            {
                T ki = ac[i + 1];
                for (int j = 0; j < i; ++j) { ki += lpc[j] * ac[i - j]; }
                ki = ki / E;
                T c = (T)1 - ki * ki;
                if (c < (T)1.0e-5) { c = (T)1.0e-5; }
                E *= c;
                tmp[i] = -ki;
                for (int j = 0; j < i; ++j) { tmp[j] = lpc[j] - ki * lpc[i - j - 1]; }
                for (int j = 0; j <= i; ++j) { lpc[j] = tmp[j]; }
            }
            for (int i = 0; i < P; ++i)
            {
                T sum = (T)0;
                for (int j = 0; j < i; ++j) { sum += (T)(i - j) * lpc[j] * cep[i - j - 1]; }
                cep[i] = -lpc[i] - sum / (T)(i + 1);
            }
            sc0[0] = floor_log(E);
        }
        __syncthreads();
        for (int k = tid; k < C; k += NTH)
        {
            const int kk = (k == C - 1 && C >= 2) ? C - 2 : k;
            T v = (kk == 0) ? sc0[0] : ((kk - 1 < P) ? cep[kk - 1] : (T)0);
            v *= p.lift[kk];
            if (p.use_energy && kk == 0) { v = M<T>::log(rawe); }
            y[k] = v;
        }
    }
}

template <typename T>
hipError_t launch_generic(const FrameParams<T> &p, hipStream_t s)
{
    if (p.ntiles == 0) { return hipSuccess; }
    const size_t lds = generic_lds_bytes(p.nfft, p.B, p.C, p.P, sizeof(T));      // above 48 KB: opted in by generic_prepare at plan creation
    if (p.nfft <= 1024) { hipLaunchKernelGGL((generic_frames_kernel<T, 64>), dim3(p.ntiles), dim3(64), lds, s, p); }
    else { hipLaunchKernelGGL((generic_frames_kernel<T, NT>), dim3(p.ntiles), dim3(NT), lds, s, p); }
    return hipGetLastError();
}

template <typename T>
hipError_t prepare_generic(int nfft, size_t lds)
{
    // the instantiation launch_generic will pick, on the current device
    const void *k = (nfft <= 1024) ? reinterpret_cast<const void *>(&generic_frames_kernel<T, 64>) : reinterpret_cast<const void *>(&generic_frames_kernel<T, NT>);
    // the attribute is the kernel's, not the plan's: opted in once per device to the largest size the host layer admits, so
    // that a later, smaller plan cannot lower what an earlier, larger plan launches with
    (void)lds;
    return opt_in_lds_once(k, (int)GENERIC_MAX_LDS);
}

}  // namespace

size_t generic_lds_bytes(int nfft, int B, int C, int P, size_t elem)
{
    (void)C;
    size_t n = (size_t)2 * nfft + NT + (size_t)B + (size_t)(B + 2) + (size_t)4 * (P + 1) + 4 + 4;
    if (nfft <= 1024 && elem == 4) { n += (size_t)2 * nfft; }                             // twiddles and window cached by the one-wave workgroups
    return (n * elem + 15) & ~(size_t)15;
}

// Opt the generic kernel that serves (nfft, elem) in to `lds` bytes of dynamic LDS on the current device.  Called when a plan is
// created, so that speech_batch_run itself only launches (and can be captured in a HIP graph).
hipError_t generic_prepare(int nfft, size_t elem, size_t lds)
{
    if (lds <= 48u * 1024u) { return hipSuccess; }
    return elem == 8 ? prepare_generic<double>(nfft, lds) : prepare_generic<float>(nfft, lds);
}

hipError_t launch_generic_f32(const FrameParams<float> &p, hipStream_t s) { return launch_generic<float>(p, s); }
hipError_t launch_generic_f64(const FrameParams<double> &p, hipStream_t s) { return launch_generic<double>(p, s); }

}  // namespace speech
