// quad_f64.hip -- the *_d entry points at 512-point frame sizes (257..512 samples: 25 ms at 16 kHz and its neighbours), in double.
//
// The generic kernel (generic_kernels.hip) spends one 64-lane workgroup and nine radix-2 passes over LDS on every frame.  This
// kernel keeps the tuned float kernels' division of labour (fast_common.hpp) in double precision: a wave owns 4 consecutive
// frames, 16 lanes per frame; the 512-point real transform is a 256-point complex one, z[m] = x[2m] + i x[2m+1],
// m = n1 + 16 n2, done as two 16-point transforms in registers around one transposition of the frame's 16 x 16 block through
// LDS (real parts, then imaginary parts); lanes k2 and 16 - k2 exchange half of their outputs and each produces the power of
// bins k and 256 - k.  The feature epilogues are the generic kernel's loops, strided over the frame's 16 lanes, so their sums run
// in the reference's order.  Nothing leaves the wave between the samples and the feature row: the only workgroup barrier is the
// one after the twiddles are parked in LDS.  Pipeline order and arithmetic: c/kaldi_mfcc.c:213-334 and :560-700 (double),
// spectrogram epilogue c/kaldi_spectrogram.c:236-251, fbank c/kaldi_fbank.c:290-350, PLP tail per SURVEY.md 8(a) row P.
//
// Not served here (the generic kernel takes them): dither, other transform sizes, B > 64 or lpc order > 32.
#include "speech_internal.hpp"
#include <cfloat>

namespace speech {

namespace {

constexpr int QT = 256;                  // threads: 4 waves x 4 frames x 16 lanes
constexpr int RS = 274;                  // doubles of a frame's region: 16 rows of 17 (transposition image), then its 257 powers

struct cd { double x, y; };
__device__ __forceinline__ cd operator+(cd a, cd b) { return {a.x + b.x, a.y + b.y}; }
__device__ __forceinline__ cd operator-(cd a, cd b) { return {a.x - b.x, a.y - b.y}; }
__device__ __forceinline__ cd cmul(cd z, double c, double s) { return {z.x * c - z.y * s, z.x * s + z.y * c}; }   // z (c + i s)
__device__ __forceinline__ cd sub_i(cd a, cd b) { return {a.x + b.y, a.y - b.x}; }                               // a - i b
__device__ __forceinline__ cd add_i(cd a, cd b) { return {a.x - b.y, a.y + b.x}; }                               // a + i b

// forward 4-point transform of (a, b, c, d) in place: outputs 0..3 land in a, b, c, d
__device__ __forceinline__ void fft4(cd &a, cd &b, cd &c, cd &d)
{
    const cd t0 = a + c, t1 = a - c, t2 = b + d, t3 = b - d;
    a = t0 + t2; c = t0 - t2;
    b = sub_i(t1, t3); d = add_i(t1, t3);
}

// forward 16-point transform in place; output k is left in z[OUT16(k)]
__host__ __device__ constexpr int OUT16(int k) { return 4 * (k & 3) + (k >> 2); }
__device__ __forceinline__ void fft16(cd (&z)[16])
{
    constexpr double C1 = 0.92387953251128675613, S1 = 0.38268343236508977173, H = 0.70710678118654752440;
#pragma unroll
    for (int j = 0; j < 4; ++j) { fft4(z[j], z[j + 4], z[j + 8], z[j + 12]); }         // z[j + 4 q] = sum_p a[j + 4 p] W4^(p q)
    // times W16^(j q)
    z[5] = cmul(z[5], C1, -S1); z[9] = cmul(z[9], H, -H); z[13] = cmul(z[13], S1, -C1);
    z[6] = cmul(z[6], H, -H); z[10] = {z[10].y, -z[10].x}; z[14] = cmul(z[14], -H, -H);
    z[7] = cmul(z[7], S1, -C1); z[11] = cmul(z[11], -H, -H); z[15] = cmul(z[15], -C1, S1);
#pragma unroll
    for (int q = 0; q < 4; ++q) { fft4(z[4 * q], z[4 * q + 1], z[4 * q + 2], z[4 * q + 3]); }   // z[4 q + r] = X[q + 4 r]
}

__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// sum over the frame's 16 lanes; every lane ends with the same bits
__device__ __forceinline__ double sum16(double v)
{
#pragma unroll
    for (int s = 8; s > 0; s >>= 1) { v += __shfl_xor(v, s, 16); }
    return v;
}

// natural log of a positive normal double (every caller floors its argument at FLT_EPSILON or passes an energy >= it): x = m 2^e with
// m in [sqrt(1/2), sqrt 2), s = (m - 1) / (m + 1), log m = 2 s (1 + s^2/3 + s^4/5 + ... + s^20/21); s^2 <= 0.0295, so the first
// dropped term is below 1e-18 of the sum.  About 35 instructions against the library routine's 80; within 2 ulp of it.
__device__ __forceinline__ double pos_log(double x)
{
    double m = __builtin_amdgcn_frexp_mant(x);                      // [0.5, 1)
    int e = __builtin_amdgcn_frexp_exp(x);
    const bool lo = m < 0.70710678118654752440;
    m = lo ? m + m : m; e = lo ? e - 1 : e;
    const double s = (m - 1.0) / (m + 1.0), z = s * s;
    double q = 1.0 / 21.0;
    q = fma(q, z, 1.0 / 19.0); q = fma(q, z, 1.0 / 17.0); q = fma(q, z, 1.0 / 15.0); q = fma(q, z, 1.0 / 13.0);
    q = fma(q, z, 1.0 / 11.0); q = fma(q, z, 1.0 / 9.0); q = fma(q, z, 1.0 / 7.0); q = fma(q, z, 1.0 / 5.0);
    q = fma(q, z, 1.0 / 3.0);
    const double two_s = s + s, tail = two_s * z * q;               // log m = 2 s + tail
    const double ed = (double)e;
    // ln 2 split so that e * LN2_HI is exact for |e| < 2^11
    constexpr double LN2_HI = 6.93147180369123816490e-01, LN2_LO = 1.90821492927058770002e-10;
    return fma(ed, LN2_HI, two_s + fma(ed, LN2_LO, tail));
}

__device__ __forceinline__ double floor_log(double v)
{
    const double eps = (double)FLT_EPSILON;
    return pos_log(v < eps ? eps : v);
}

// acc + sum of a[i] * b[i], i = 0..n-1, added in index order; the operands are fetched eight at a time so that the chain of
// additions waits for memory once per eight terms, not once per term
__device__ __forceinline__ double dot_in_order(double acc, const double *a, const double *b, int n)
{
    int i = 0;
    for (; i + 8 <= n; i += 8)
    {
        double u[8], w[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) { u[r] = a[i + r]; w[r] = b[i + r]; }
#pragma unroll
        for (int r = 0; r < 8; ++r) { acc += u[r] * w[r]; }
    }
    for (; i < n; ++i) { acc += a[i] * b[i]; }
    return acc;
}

template <bool PCM>
__global__ __launch_bounds__(QT) void quad_frames_f64_kernel(const FrameParams<double> p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int tid = threadIdx.x;
    const int L = p.L, F = p.F, B = p.B, C = p.C, P = p.P;
    double *c_twr = reinterpret_cast<double *>(smem_raw);          // exp(-2 pi i k / 512), k < 256
    double *c_twi = c_twr + 256;
    const int per = RS + ((p.kind == SPEECH_SPECTROGRAM) ? 0 : (2 * B + 4 + 4 * (P + 1)));
    const int fr = tid >> 4, n1 = tid & 15;
    double *reg = c_twi + 256 + (size_t)fr * per;                    // the frame's region
    double *pw = reg;
    double *mel = reg + RS;                                          // B
    double *dup = mel + B;                                           // B + 2
    double *ac = dup + (B + 2);                                      // P + 1
    double *lpc = ac + (P + 1), *tmp = lpc + (P + 1), *cep = tmp + (P + 1);
    double *sc0 = cep + (P + 1);                                     // 1

    c_twr[tid] = p.tw_re[tid]; c_twi[tid] = p.tw_im[tid];
    __syncthreads();

    const TileDesc tile = p.tiles[blockIdx.x];
    if ((uint32_t)(fr & ~3) >= tile.nframes) { return; }             // (wave-uniform) none of this wave's four frames exists
    const bool live = (uint32_t)fr < tile.nframes;                   // a missing frame of a live wave recomputes the tile's last one and stores nothing
    const UttDesc utt = p.utts[tile.utt];
    const long long N = (long long)utt.n;
    const double eps = (double)FLT_EPSILON;
    const long long w = (long long)tile.frame0 + (live ? fr : (int)tile.nframes - 1);
    double *y = p.Y + utt.y_off + (unsigned long long)w * (unsigned long long)p.D;
    // frame start: c/kaldi_mfcc.c:81-82,216-240
    const long long start = p.snip_edges ? w * p.stp : (long long)(p.stp / 2) - (long long)(L / 2) + w * p.stp;
    const bool interior = (start >= 0 && start + L <= N);

    // lane n1 holds samples 2 n1 + 32 n2 (+ 1), n2 = 0..15, and for the pre-emphasis the sample before its even one (pv);
    // rows 0..7 lie inside any frame this kernel serves (L > 256)
    cd z[16];
    double pv[16];
    if (__builtin_amdgcn_ballot_w64(!interior) == 0)                 // (wave-uniform) no reflection in any of the four frames: plain offsets
    {
        const long long at = (long long)utt.x_off + start + 2 * n1;
        const double *xf = p.X + at;
        const short *xs = p.X16 + at;
        const int back = (n1 > 0) ? -1 : 0;                          // sample 0 of a frame stands in for its own predecessor (c/kaldi_mfcc.c:288)
#pragma unroll
        for (int n2 = 0; n2 < 16; ++n2)
        {
            const int l0 = 2 * n1 + 32 * n2;
            const bool ia = (n2 < 8) || (l0 < L), ib = (n2 < 8) || (l0 + 1 < L);
            z[n2].x = ia ? (PCM ? (double)xs[32 * n2] : xf[32 * n2]) : 0.0;
            z[n2].y = ib ? (PCM ? (double)xs[32 * n2 + 1] : xf[32 * n2 + 1]) : 0.0;
            const int pb = (n2 == 0) ? back : 32 * n2 - 1;
            pv[n2] = (p.do_preemph && ia) ? (PCM ? (double)xs[pb] : xf[pb]) : 0.0;
        }
    }
    else
    {
        const double *x = p.X + utt.x_off;
        const short *x16 = p.X16 + utt.x_off;
        // sample l of the frame (l < L), reflected at the utterance's ends: c/kaldi_mfcc.c:228
        auto sample = [&](int l) -> double {
            long long n = start + l;
            while (n < 0 || n >= N) { n = (n < 0) ? -n - 1 : 2 * N - 1 - n; }
            return PCM ? (double)x16[n] : x[n];
        };
        for (int n2 = 0; n2 < 16; ++n2)
        {
            const int l0 = 2 * n1 + 32 * n2;
            const double a = (l0 < L) ? sample(l0) : 0.0, b = (l0 + 1 < L) ? sample(l0 + 1) : 0.0;
            const double pr = (l0 < L) ? sample(l0 > 0 ? l0 - 1 : 0) : 0.0;
#pragma unroll
            for (int r = 0; r < 16; ++r) { if (r == n2) { z[r].x = a; z[r].y = b; pv[r] = pr; } }     // registers, not a private array
        }
    }
    double rawe = 0.0;
    auto energy = [&]() -> double {
        double s = 0.0;
#pragma unroll
        for (int n2 = 0; n2 < 16; ++n2) { s += z[n2].x * z[n2].x; s += z[n2].y * z[n2].y; }
        s = sum16(s);
        return s < eps ? eps : s;
    };
    if (p.want_pre && p.pre_before_dc) { rawe = energy(); }          // kaldi_mfcc_d only: c/kaldi_mfcc.c:576-582
    double mn = 0.0;
    if (p.dc0)                                                       // c/kaldi_mfcc.c:265-271
    {
        double s = 0.0;
#pragma unroll
        for (int n2 = 0; n2 < 16; ++n2) { s += z[n2].x; s += z[n2].y; }
        mn = sum16(s) / (double)L;
#pragma unroll
        for (int n2 = 0; n2 < 16; ++n2)
        {
            const int l0 = 2 * n1 + 32 * n2;
            if (l0 < L) { z[n2].x -= mn; }
            if (l0 + 1 < L) { z[n2].y -= mn; }
        }
    }
    if (p.want_pre && !p.pre_before_dc) { rawe = energy(); }         // c/kaldi_mfcc.c:274-279
    // pre-emphasis (closed form of the backwards in-place loop, c/kaldi_mfcc.c:282-288) and window (:291)
#pragma unroll
    for (int n2 = 0; n2 < 16; ++n2)
    {
        const int l0 = 2 * n1 + 32 * n2;
        double a = z[n2].x, b = z[n2].y;
        if (p.do_preemph)
        {
            const double prev = pv[n2] - mn;                         // mn is 0 without dc0; sample 0 meets itself
            b -= p.preemph * a;                                      // sample l0 + 1 takes sample l0 (zero beyond L either way)
            a -= p.preemph * prev;
        }
        z[n2].x = (l0 < L) ? a * p.win[l0] : 0.0;
        z[n2].y = (l0 + 1 < L) ? b * p.win[l0 + 1] : 0.0;
    }
    if (p.want_post) { rawe = energy(); }                            // c/kaldi_mfcc.c:295-301

    // 256-point complex transform (replaces the FFTW R2HC call, c/kaldi_mfcc.c:304)
    fft16(z);                                                        // over n2: output k2 in z[OUT16(k2)]
    cd v[16];
#pragma unroll
    for (int k2 = 0; k2 < 16; ++k2)                                  // times W256^(n1 k2) = tw[2 n1 k2], sign flipped past half a turn
    {
        const int j = n1 * k2;
        const double sg = (j & 128) ? -1.0 : 1.0;
        const double c = sg * c_twr[(2 * j) & 255], s = sg * c_twi[(2 * j) & 255];
        v[k2] = (k2 == 0) ? z[OUT16(0)] : cmul(z[OUT16(k2)], c, s);
    }
    wave_sync();
#pragma unroll
    for (int k2 = 0; k2 < 16; ++k2) { reg[k2 * 17 + n1] = v[k2].x; }
    wave_sync();
#pragma unroll
    for (int m = 0; m < 16; ++m) { z[m].x = reg[n1 * 17 + m]; }       // lane k2 = n1 now holds Y[m][k2], m = the old n1
    wave_sync();
#pragma unroll
    for (int k2 = 0; k2 < 16; ++k2) { reg[k2 * 17 + n1] = v[k2].y; }
    wave_sync();
#pragma unroll
    for (int m = 0; m < 16; ++m) { z[m].y = reg[n1 * 17 + m]; }
    wave_sync();
    fft16(z);                                                        // Z[k2 + 16 k1] in z[OUT16(k1)]

    // untangle + power, c/kaldi_mfcc.c:307-310: with A = Z[k] + conj Z[256-k], B = Z[k] - conj Z[256-k] and w = exp(-2 pi i k / 512),
    // X[k] = (A - i w B) / 2 and X[256-k] = conj(A + i w B) / 2.  Lane k2 does k = k2 + 16 k1 for k1 = 0..7; Z[256-k] is
    // register 15 - k1 of lane 16 - k2 (lane 0: its own register 16 - k1).  X[0] = Zr[0] + Zi[0] and X[256] = Zr[0] - Zi[0].
    const int k2 = n1, partner = (tid & 48) | ((16 - k2) & 15);          // lane number within the wave
#pragma unroll
    for (int k1 = 0; k1 < 8; ++k1)
    {
        const cd send = z[OUT16(15 - k1)];
        cd q;
        q.x = __shfl(send.x, partner, 64); q.y = __shfl(send.y, partner, 64);
        if (k2 == 0) { q = (k1 == 0) ? z[OUT16(0)] : z[OUT16(16 - k1)]; }
        const cd pz = z[OUT16(k1)];
        const int k = k2 + 16 * k1;
        const double Ar = pz.x + q.x, Ai = pz.y - q.y, Br = pz.x - q.x, Bi = pz.y + q.y;
        const double c = c_twr[k], s = c_twi[k];
        const double ur = c * Bi + s * Br, ui = c * Br - s * Bi;
        const double xr = Ar + ur, xi = Ai - ui, yr = Ar - ur, yi = Ai + ui;
        double pa = 0.25 * (xr * xr + xi * xi), pb = 0.25 * (yr * yr + yi * yi);
        if (k == 0) { const double x0 = pz.x + pz.y, xn = pz.x - pz.y; pa = x0 * x0; pb = xn * xn; }
        pw[k] = pa; pw[256 - k] = pb;
    }
    if (k2 == 0)                                                     // bin 128 pairs with itself
    {
        const cd pz = z[OUT16(8)];
        const double Ar = 2.0 * pz.x, Bi = 2.0 * pz.y;              // A = Z + conj Z, B = Z - conj Z
        const double c = c_twr[128], s = c_twi[128];
        const double xr = Ar + c * Bi, xi = s * Bi;                  // Ai = Br = 0
        pw[128] = 0.25 * (xr * xr + xi * xi);
    }
    wave_sync();

    if (p.kind == SPEECH_SPECTROGRAM)
    {
        // c/kaldi_spectrogram.c:236-251
        if (live) { for (int k = n1; k < F; k += 16) { y[k] = floor_log((k == 0) ? rawe : ((k == F - 1) ? pw[F - 2] : pw[k])); } }
        return;
    }

    // sparse mel sums, c/kaldi_mfcc.c:313-320 / c/kaldi_fbank.c:312-345
    for (int b = n1; b < B; b += 16)
    {
        const double *wgt = p.f2b + p.mel_off[b];
        const int beg = p.mel_beg[b], num = p.mel_num[b];
        double sm = 0.0;
        if (p.amp) { for (int f = 0; f < num; ++f) { sm += ::sqrt(pw[beg + f]) * wgt[f]; } }
        else { sm = dot_in_order(0.0, pw + beg, wgt, num); }
        mel[b] = sm;
    }
    wave_sync();

    if (p.kind == SPEECH_FBANK)
    {
        const int eo = p.use_energy ? 1 : 0;
        if (!live) { return; }
        for (int b = n1; b < B; b += 16)
        {
            const int src = (b == B - 1 && B >= 2) ? B - 2 : b;     // c/kaldi_fbank.c:350
            const double m = mel[src];
            double out = p.lg ? floor_log(m) : m;
            if (B == 1 && eo) { out = pos_log(rawe); }
            y[eo + b] = out;
        }
        if (eo && n1 == 0) { y[0] = pos_log(rawe); }                   // c/kaldi_fbank.c:299
        return;
    }

    if (p.kind == SPEECH_MFCC)
    {
        for (int b = n1; b < B; b += 16) { mel[b] = floor_log(mel[b]); }    // c/kaldi_mfcc.c:318
        wave_sync();
        if (!live) { return; }
        for (int c = n1; c < C; c += 16)
        {
            const int cc = (c == C - 1 && C >= 2) ? C - 2 : c;      // c/kaldi_mfcc.c:333
            const double *d = p.dct + (size_t)cc * B;
            const double acc = dot_in_order(0.0, mel, d, B);
            double out = p.lift[cc] * acc;                           // c/kaldi_mfcc.c:327-329
            if (p.use_energy && cc == 0) { out = pos_log(rawe); }
            y[c] = out;
        }
        return;
    }

    // PLP tail
    for (int j = n1; j < B + 2; j += 16)
    {
        const int b = (j == 0) ? 0 : ((j == B + 1) ? B - 1 : j - 1);
        const double m = mel[b] * p.eql[b];
        dup[j] = (p.compress_mode == 1) ? m : ((p.compress_mode == 2) ? ::sqrt(m) : ::pow(m, p.compress));
    }
    wave_sync();
    for (int i = n1; i <= P; i += 16)
    {
        const double *row = p.idft + (size_t)i * (B + 2);
        ac[i] = dot_in_order(0.0, row, dup, B + 2);
    }
    wave_sync();
    if (n1 == 0)
    {
        double E = ac[0];
        for (int i = 0; i < P; ++i)
        {
            double ki = ac[i + 1];
            for (int j = 0; j < i; ++j) { ki += lpc[j] * ac[i - j]; }
            ki = ki / E;
            double c = 1.0 - ki * ki;
            if (c < 1.0e-5) { c = 1.0e-5; }
            E *= c;
            tmp[i] = -ki;
            for (int j = 0; j < i; ++j) { tmp[j] = lpc[j] - ki * lpc[i - j - 1]; }
            for (int j = 0; j <= i; ++j) { lpc[j] = tmp[j]; }
        }
        for (int i = 0; i < P; ++i)
        {
            double sum = 0.0;
            for (int j = 0; j < i; ++j) { sum += (double)(i - j) * lpc[j] * cep[i - j - 1]; }
            cep[i] = -lpc[i] - sum / (double)(i + 1);
        }
        sc0[0] = floor_log(E);
    }
    wave_sync();
    if (!live) { return; }
    for (int k = n1; k < C; k += 16)
    {
        const int kk = (k == C - 1 && C >= 2) ? C - 2 : k;
        double out = (kk == 0) ? sc0[0] : ((kk - 1 < P) ? cep[kk - 1] : 0.0);
        out *= p.lift[kk];
        if (p.use_energy && kk == 0) { out = pos_log(rawe); }
        y[k] = out;
    }
}

}  // namespace

size_t quad_f64_lds_bytes(int kind, int B, int P)
{
    const size_t per = RS + ((kind == SPEECH_SPECTROGRAM) ? 0 : (2 * (size_t)B + 4 + 4 * ((size_t)P + 1)));
    return (512 + QUAD_F64_TILE_FRAMES * per) * sizeof(double);
}

// does this option set run on the kernel above?  (`P` = the plan's lpc order, 0 unless plp)
bool quad_f64_supported(const speech_opts &o, size_t nfft, size_t P)
{
    if (o.dtype != SPEECH_F64 || o.force_generic || nfft != 512u) { return false; }
    if (o.dither > (double)FLT_EPSILON) { return false; }
    if (o.kind != SPEECH_SPECTROGRAM && (o.B > 64u || P > 32u)) { return false; }
    return quad_f64_lds_bytes(o.kind, o.kind == SPEECH_SPECTROGRAM ? 0 : (int)o.B, (int)P) <= 64u * 1024u;
}

hipError_t launch_quad_f64(const FrameParams<double> &p, hipStream_t s)
{
    if (p.ntiles == 0) { return hipSuccess; }
    if (p.nfft != 512 || p.L <= 256 || p.L > 512 || p.do_dither) { return hipErrorInvalidValue; }   // what the indexing above assumes
    const size_t lds = quad_f64_lds_bytes(p.kind, p.B, p.P);
    if (p.X16) { hipLaunchKernelGGL(quad_frames_f64_kernel<true>, dim3(p.ntiles), dim3(QT), lds, s, p); }
    else { hipLaunchKernelGGL(quad_frames_f64_kernel<false>, dim3(p.ntiles), dim3(QT), lds, s, p); }
    return hipGetLastError();
}

}  // namespace speech
