// speech_internal.hpp -- structures shared by the host layer and the kernel translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>
#include "../../include/speech_hip.h"

namespace speech {

// One utterance of a batch (element offsets into the packed device arrays).
struct UttDesc {
    unsigned long long x_off;   // first sample
    unsigned long long n;       // samples
    unsigned long long y_off;   // first output element
    unsigned long long W;       // frames
};

// One workgroup's run of consecutive frames of one utterance.
struct TileDesc {
    uint32_t utt;
    uint32_t frame0;
    uint32_t nframes;
    uint32_t uid;               // the utterance's number for the dither generator: its index, or what speech_batch_set_ids gave
};

// Dither (c/kaldi_mfcc.c:243-262: a Box-Muller pair per two samples of every frame, drawn afresh for each frame).  The reference
// seeds a serial generator from the wall clock; here the pair of samples (2i, 2i+1) of frame w of utterance u takes its
// random bits from a counter-based hash, so any kernel shape draws the same noise:
//   fk = mix64(mix64(seed ^ uid << 32) + w * 0x100000001B3)            once per frame (64-bit, well mixed)
//   h  = fmix32(lo32(fk) ^ i * 0x9E3779B9) + hi32(fk)                per pair: 32 bits, 8 instructions
//   u1 = hi16(h) / 2^16,  u2 = lo16(h) / 2^16                        (65536 levels each: |noise| <= 4.7 sigma, ample for dither)
#if defined(__HIPCC__)
__host__ __device__ __forceinline__ unsigned long long mix64(unsigned long long z)
{
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
constexpr unsigned long long DITHER_FRAME_MUL = 0x100000001B3ull;

// key of frame w of an utterance whose key is ku = mix64(seed ^ u << 32)
__host__ __device__ __forceinline__ unsigned long long dither_frame_key(unsigned long long ku, unsigned long long w) { return mix64(ku + w * DITHER_FRAME_MUL); }

// the 32 random bits of pair i of a frame (murmur3's 32-bit finaliser on the pair counter)
__host__ __device__ __forceinline__ unsigned dither_bits(unsigned long long fk, unsigned i)
{
    unsigned h = (unsigned)fk ^ (i * 0x9E3779B9u);
    h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
    return h + (unsigned)(fk >> 32);
}

// float: hardware log2 / sqrt / sin / cos (arguments in revolutions) -- it is noise; what matters is that the tuned and the
// generic kernels compute it with the same instructions.  g0 goes to the even sample, g1 to the odd one.
__device__ __forceinline__ void dither_pair(unsigned h, float scale, float &g0, float &g1)
{
    const float u1 = (float)(h >> 16) * 1.52587890625e-05f;                  // < 1: 1 - u1 >= 2^-16
    const float u2 = (float)(h & 0xFFFFu) * 1.52587890625e-05f;
    const float R = scale * __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(1.0f - u1));     // sqrt(-2 ln(1-u1))
    g0 = R * __builtin_amdgcn_sinf(u2 + 0.25f);                              // cos, as the sine a quarter turn on (exact sum): one
    g1 = R * __builtin_amdgcn_sinf(u2);                                      // code path for dither_one below
}
// one member of the pair: odd = 0 the even sample's (g0), 1 the odd sample's (g1); the same bits as dither_pair gives
__device__ __forceinline__ float dither_one(unsigned h, float scale, int odd)
{
    const float u1 = (float)(h >> 16) * 1.52587890625e-05f;
    const float u2 = (float)(h & 0xFFFFu) * 1.52587890625e-05f;
    const float R = scale * __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(1.0f - u1));
    return R * __builtin_amdgcn_sinf(u2 + (odd ? 0.0f : 0.25f));
}
// Synthetic corpus audio: integer arithmetic only, so any machine produces the same waveform.  Sample n of
// corpus utterance `utt` under master seed `seed`:
//   ku    = mix64(seed ^ utt)
//   noise = (sum of the four 14-bit fields of mix64(ku ^ n * 0xD6E8FEB86659FD93) - 32766) >> 3      near-Gaussian, ~ -29 dBFS
//   tone k (k = 0, 1, 2), from r_k = mix64(ku + k + 1): a triangle wave on a 32-bit phase accumulator,
//           phase = lo32(r_k) + n * inc_k,  inc_k = 0x0147AE14 + (hi32(r_k) * 0x71EB851F >> 32)       80 .. 7200 Hz at 16 kHz
//           t = phase >> 16;  tri = t < 32768 ? t : 65535 - t;  tone = (tri - 16384) * amp_k >> 14,  amp_k = 500 + (r_k >> 20 & 2047)
//   sample = noise + tone_0 + tone_1 + tone_2        (|sample| < 12 000: int16 scale, as Kaldi waveforms)
// speech_amd/corpus.py holds the numpy statement of the same lines; tests compare the two bit for bit.
struct SynthUtt { unsigned long long ku; unsigned ph[3], inc[3]; int amp[3]; };
__host__ __device__ __forceinline__ SynthUtt synth_utt(unsigned long long seed, unsigned long long utt)
{
    SynthUtt s;
    s.ku = mix64(seed ^ utt);
    for (int k = 0; k < 3; ++k)
    {
        const unsigned long long r = mix64(s.ku + (unsigned long long)(k + 1));
        s.ph[k] = (unsigned)r;
        s.inc[k] = 0x0147AE14u + (unsigned)(((r >> 32) * 0x71EB851Full) >> 32);
        s.amp[k] = 500 + (int)((r >> 20) & 2047ull);
    }
    return s;
}
__host__ __device__ __forceinline__ int synth_sample(const SynthUtt &s, unsigned n)
{
    const unsigned long long h = mix64(s.ku ^ ((unsigned long long)n * 0xD6E8FEB86659FD93ull));
    int v = (int)((h & 0x3FFFull) + ((h >> 14) & 0x3FFFull) + ((h >> 28) & 0x3FFFull) + ((h >> 42) & 0x3FFFull)) - 32766;
    v >>= 3;                                                                 // arithmetic: floor
    for (int k = 0; k < 3; ++k)
    {
        const unsigned t = (s.ph[k] + n * s.inc[k]) >> 16;
        const int tri = (int)(t < 32768u ? t : 65535u - t);
        v += ((tri - 16384) * s.amp[k]) >> 14;
    }
    return v;
}

__device__ __forceinline__ void dither_pair(unsigned h, double scale, double &g0, double &g1)
{
    const double u1 = (double)(h >> 16) * 1.52587890625e-05;
    const double u2 = (double)(h & 0xFFFFu) * 1.52587890625e-05;
    const double R = scale * ::sqrt(-2.0 * ::log(1.0 - u1));
    const double ang = 6.283185307179586476925286766559 * u2;
    g0 = R * ::cos(ang);
    g1 = R * ::sin(ang);
}
#endif

// Everything a per-frame kernel needs besides the data (device pointers are to the plan's tables).
template <typename T>
struct FrameParams {
    const T *X;
    const short *X16;            // non-null: 16-bit PCM input (same element offsets), X is then unused
    T *Y;
    const UttDesc *utts;
    const TileDesc *tiles;
    uint32_t ntiles;
    uint32_t nutt, max_frames;   // utterances of the batch, frames of its longest one (grids over (utterance, chunk of frames))
    T *plp_en;                   // tuned plp with use_energy: one float per output row, kept between its two launches (else null)
    int kind;
    int L, stp, nfft, F, log2n;
    int D;                       // feature dimension
    int snip_edges, dc0, raw_energy, use_energy, amp, lg;
    int want_pre, want_post, pre_before_dc;
    int do_preemph, do_dither;
    T preemph, dither;
    unsigned long long seed;
    const T *win;                // [L]
    const T *tw_re, *tw_im;      // generic FFT twiddles exp(-2 pi i k/nfft), k < nfft/2
    // mel (sparse, as c/kaldi_mfcc.c:157-188)
    int B;
    const T *f2b;                // concatenated non-zero runs
    const int *mel_beg, *mel_num, *mel_off;
    // mfcc
    int C;
    const T *dct;                // [C x B] 2 cos(pi (b+1/2) c / B)
    const T *lift;               // [C]
    // plp
    int P;                       // LPC order
    T compress;
    int compress_mode;           // 0 pow, 1 identity, 2 sqrt
    const T *eql;                // [B]
    const T *idft;               // [(P+1) x (B+2)]
};

// host layer (speech_api.hip), for host_pipeline.hip
int fail(const char *fn, const char *fmt, ...);                             // records and prints "error in <fn>: ...", returns 1
void plan_info(const speech_plan *p, int *device, int *dtype, size_t *D);
int batch_fill(speech_batch *b, speech_plan *p, size_t nutt, const size_t *n_samples, const size_t *x_off, const size_t *y_off, uint32_t uid_base,
               const unsigned int *uids);
int batch_create(speech_batch **out, speech_plan *p, size_t nutt, const size_t *n_samples, const size_t *x_off, const size_t *y_off, uint32_t uid_base,
                 const unsigned int *uids);

// corpus post-processing (post_kernels.hip)
constexpr int POST_MAX_ORDER = 3, POST_MAX_WINDOW = 5, POST_MAX_TAPS = 2 * POST_MAX_ORDER * POST_MAX_WINDOW + 1;
enum { POST_MOMENTS = 0, POST_AFFINE = 1, POST_DELTAS = 2, POST_SPLICE = 3, POST_SPLICE_SUB = 4 };
struct PostParams {
    void *Y;                     // in place (affine) or output (deltas, splice)
    const void *Yin;             // input (moments, deltas, splice)
    const UttDesc *utts;
    uint32_t nutt;
    int D, rows_per_wg;
    double *mom;                 // [nutt][2][D]: column sums and sums of squares per utterance
    const double *scale, *offset;    // affine mode 0: [D] each (device)
    int mode, norm_vars;
    int order, window, left, right;
    int vec16;                       // splice: D, every utterance's offset and both arrays allow 16-byte units
    int sub_n, sub_off;              // splice + subsample: output row j of an utterance is spliced frame sub_off + j sub_n
    const unsigned long long *zrow;  // [nutt] first output row of each utterance in the dense packing
    double dscale[POST_MAX_ORDER + 1][POST_MAX_TAPS];    // delta filters, centred: order k has 2 k window + 1 taps
};
hipError_t launch_post(int what, bool f64, const PostParams &p, uint32_t max_frames, hipStream_t s);
constexpr int POST_REDUCE_GROUP = 64;
// out[g][c] = sum of in[64 g .. 64 g + 63][c] in row order; out holds ceil(n / 64) rows
hipError_t launch_reduce_rows(const double *in, double *out, size_t n, int cols, hipStream_t s);

// synthetic corpus audio (synth_kernels.hip): elem = bytes per sample of X (2: int16, 4: float, 8: double)
hipError_t launch_synth(void *X, int elem, const UttDesc *utts, const unsigned long long *ids, uint32_t nutt, unsigned long long max_n,
                        unsigned long long seed, hipStream_t s);

// launchers (defined in the kernel translation units)
hipError_t launch_generic_f32(const FrameParams<float> &p, hipStream_t s);
hipError_t launch_generic_f64(const FrameParams<double> &p, hipStream_t s);
size_t generic_lds_bytes(int nfft, int B, int C, int P, size_t elem);
hipError_t generic_prepare(int nfft, size_t elem, size_t lds);
constexpr size_t GENERIC_MAX_LDS = (size_t)160 * 1024;                      // what speech_plan_create admits for a generic plan: a CU's LDS
hipError_t opt_in_lds_once(const void *kernel, int bytes);                  // fast_kernels.hip: once per (device, kernel), always to the kernel's largest admitted size
hipError_t post_prepare(bool f64, int D);                                  // the affine pass keeps 2 D scale / offset values in LDS

// double precision at 512-point frame sizes (quad_f64.hip): 4 frames per wave, 16 per workgroup
constexpr int QUAD_F64_TILE_FRAMES = 16;
bool quad_f64_supported(const speech_opts &o, size_t nfft, size_t P);
size_t quad_f64_lds_bytes(int kind, int B, int P);
hipError_t launch_quad_f64(const FrameParams<double> &p, hipStream_t s);

constexpr int GENERIC_TILE_FRAMES = 8;
constexpr int GENERIC_MAX_NFFT = 8192;

}  // namespace speech
