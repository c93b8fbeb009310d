// fast_common.hpp -- shared by the tuned gfx950 kernels (fast_spectrogram.hip, fast_mel.hip) and their host side
// (fast_kernels.hip): plan tables and launch parameters, and the device code of one frame's front end + FFT.
//
// Unit of work: a QUAD = 4 consecutive frames of one utterance, owned by ONE 64-lane wavefront
// (16 lanes per frame).  Mel kinds: a 512-thread workgroup takes a tile of 32 consecutive frames, one quad per wave;
// spectrogram: 256 threads walk a tile of 64 frames in four sub-tiles.
//
//   audio      the wave loads the quad's contiguous sample span ((4-1)*stp + L floats) with coalesced
//              global loads into its private LDS slice; the overlapping frames are cut from LDS
//              (reference: gather, c/kaldi_mfcc.c:216-240).
//   front end  DC removal, raw energy, pre-emphasis, window in registers; the two sums are DPP
//              reductions over the frame's 16 lanes (reference: c/kaldi_mfcc.c:265-301).
//   FFT        512-point real FFT as a 256-point complex FFT of z[n] = x[2n] + i x[2n+1], n = n1 + 16 n2:
//              lane n1 does a 16-point FFT over n2 in registers, multiplies by W256^(n1 k2), the frame's
//              16x16 block is transposed through the wave's LDS slice (XOR-swizzled; mel kinds: real parts, then
//              imaginary parts, to keep the slice at 4 KB; spectrogram: complex pairs), lane k2 does the second
//              16-point FFT over n1.  No workgroup barrier (replaces FFTW R2HC, c/kaldi_mfcc.c:304).
//   untangle   lanes k2 and 16-k2 swap half of their outputs (ds_bpermute) and each computes the power of
//              bins k and 256-k for 8 values of k (reference: power, c/kaldi_mfcc.c:307-310).
//   epilogue   spectrogram: log / energy / Nyquist quirk (c/kaldi_spectrogram.c:236-251), staged in LDS so
//              the quad's 4 x 257 outputs leave as one contiguous coalesced run.
//              fbank / mfcc / plp: the tile's 32 x 257 power block is staged in LDS and contracted with the
//              banded mel matrix on the matrix cores (v_mfma_f32_16x16x4_f32, exact f32 fma chains); MFCC
//              adds a second MFMA contraction with the lifter-scaled DCT matrix; PLP an MFMA with the IDFT
//              bases followed by one lane per frame running Levinson-Durbin and the cepstral recursion.
#pragma once
#include "speech_internal.hpp"
#include <cfloat>
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>

namespace speech {

constexpr int MAX_MTILES = 8;            // mel tiles of up to 16 columns (B <= 128), balanced over the waves; cepstral tiles of 16 (C <= 96)

struct FastGeom { bool ok, pair, wide, xwide; int fw, nz, tile; size_t span, ws, lds, m2; };  // m2: floats of mat2 staged in LDS (0: read from global)      // launch geometry of a plan, fast_geom()

struct FastTables {
    float *d_tab;      // win[512] (zero beyond L) | tw1[16][16] complex | tw2[8][16] complex
    float *d_melw;     // banded mel matrix: per 16-mel tile, [4 * ksteps][16] weights of FFT bins kbeg .. kbeg + 4 ksteps - 1
    float *d_mat2;     // mfcc: lifter-scaled DCT matrix [Bp4][Cp]; plp: IDFT bases [K4][16]
    float *d_aux;      // plp: eql[B] | lift[C]
    int nz;            // rows of samples that hold data: ceil(L / 32), pair mode ceil(L / 16)
    int F;             // bins: 257, or 129 for 256-point plans
    FastGeom geom;
    int kind, B, C, P, D, MP, Cp, K2, ntile_m, ntile_c;
    int mel_kbeg[MAX_MTILES], mel_ksteps[MAX_MTILES], mel_woff[MAX_MTILES], mel_first[MAX_MTILES], mel_cnt[MAX_MTILES];
};
void fast_tables_destroy(FastTables *t);


constexpr int NT = 256;                 // threads per workgroup (4 waves)
constexpr int QF = 4;                   // frames per wave (quad)
constexpr int MW = 8;                   // waves per workgroup, mel kinds: with 32 frames the matrix-core phases have two row tiles
                                        // per mel tile, i.e. a job for every wave, and half as many workgroup barriers per frame
constexpr int NTM = 64 * MW;            // threads per workgroup, mel kinds

constexpr int TILE_SPEC = 64;  // frames per workgroup, spectrogram: 4 sub-tiles of 16 frames, one quad per wave per sub-tile
constexpr int NPRE = 16;                // prefetch registers per lane: the next quad's audio (up to 1024 samples)
constexpr int TR_WAVE_C = 2048;         // floats of a wave's slice when the transposition image is complex (spectrogram kernel)
constexpr int NFFT = 512, NC = 256;
constexpr int TAB_WIN = 0, TAB_TW1 = 512, TAB_TW2 = 1024, TAB_FLOATS = 1024 + 8 * 16 * 2;
constexpr int TR_WAVE = 1032;           // floats of a wave's LDS slice: 4 frames x 16 x 16 reals (transposition image of the real, then the
                                        // imaginary parts), 4 x 257 outputs; four slices = the 16 x 258 power tile
constexpr int PP2 = 130;                // pitch of the 64 x 129 power tile of the pair mode (also = 2 mod 32)
constexpr int PP = 258;                 // pitch of the TILE x 257 power tile (= 2 mod 32: conflict-free MFMA operand reads)
// 1024-point plans ("wide": one frame on 32 lanes, frame_power_wide): win[1024] | tw1 | tw2 | tw3[8][16] complex = W1024^(k2 + 16 k1)
constexpr int TABW_WIN = 0, TABW_TW1 = 1024, TABW_TW2 = 1536, TABW_TW3 = 1792, TABW_FLOATS = 2048;
constexpr int PPW = 514;                // pitch of the 16 x 513 power tile of the wide mode (= 2 mod 32)
// 2048-point plans ("xwide": one frame on the wave's 64 lanes, frame_power_xwide): tw1 | tw2 | tw3[9][16] complex = W1024^(k2 + 16 k1)
// | tw4[9][16] complex = W2048^(k2 + 16 k1) (row k1 = 8 serves bin 128 of lane k2 = 0) | win[2048, padded].
// A lane of that mode reads samples (and window factors) 8 apart, which in a dense array puts 32 lanes on 8 banks; the
// audio slice and the window table therefore carry 2 floats of padding after every 32 (sample i lives at padx(i)):
// 8 a + 2 b + g (a, b = 0..3, g = 0..1) are 32 different banks, and the rows of 64 samples the loads park stay conflict-free.
__host__ __device__ constexpr int padx(int i) { return i + 2 * (i >> 5); }
// The window comes LAST: the mel kernels stage only the twiddles (TABX_TWF floats) in LDS and read the window through L1 (8.7 KB
// less LDS per workgroup is what lets three of them share a CU); the spectrogram kernel stages the whole table.
constexpr int TABX_TW1 = 0, TABX_TW2 = 512, TABX_TW3 = 768, TABX_TW4 = 1056, TABX_TWF = 1344, TABX_WIN = 1344, TABX_FLOATS = 3520;
constexpr int PPX = 1026;               // pitch of the 8 x 1025 power tile of the xwide mode (= 2 mod 32)
constexpr unsigned BUF_FLAGS = 0x00020000u;   // word 3 of a raw buffer descriptor on gfx9/CDNA: 32-bit data format

struct FastParams {
    const float *X;
    const short *X16;                   // non-null: the audio is 16-bit PCM (speech_batch_run_pcm16), converted on the way into LDS
    float *Y;
    const UttDesc *utts;
    const TileDesc *tiles;
    uint32_t ntiles;
    int L, stp, D, span, ws;            // span: floats of a quad's audio, rounded up to 64; ws: floats of a wave's LDS slice (>= span, TR_WAVE)
    int snip_edges, dc0, want_pre, want_post;
    float preemph;                      // 0 when pre-emphasis is off
    float invL;                         // 1 / L: the frame mean is sum * invL (an IEEE division is 10 instructions; the lanes' tree sum already rounds otherwise than the reference's running sum)
    int do_dither;                      // c/kaldi_mfcc.c:243: dither > FLT_EPSILON
    float dither;
    unsigned long long seed;
#ifdef SPEECH_DIAG
    int ablate;                         // libspeech_diag.so only (make diag; SPEECH_ABLATE): 1 = no feature stores, 2 = no audio loads, 4 = mel kinds stop after the FFT, 8 = stores land in a 2 MB (L2-resident) window, 16 = every utterance reads utterance 0's audio, 32 / 64 = mel kinds stop after the power tile / after the mel contraction, 256 = mel kinds load their audio through registers (no buffer_load ... lds)
#endif
    const float *tab;
    // mel kinds
    const float *melw, *mat2, *aux;
    float *plp_en;                      // plp with use_energy: the batch's energy scratch, one float per output row (row = y_off / C + frame)
    int B, C, MP, Cp, K2, ntile_m, ntile_c;
    int m2;                             // floats of mat2 staged in LDS between the tables and the slices (0: not staged)
    int amp, lg, use_energy, compress_mode;
    float compress;
    int mel_kbeg[MAX_MTILES], mel_ksteps[MAX_MTILES], mel_woff[MAX_MTILES], mel_first[MAX_MTILES], mel_cnt[MAX_MTILES];
};

// Timing diagnostics exist only in libspeech_diag.so (`make diag`, -DSPEECH_DIAG): the release library has no
// such field, reads no environment variable on the launch path and its kernels carry none of the switches.
#ifdef SPEECH_DIAG
#define SPEECH_ABL(p) ((p).ablate)
#else
#define SPEECH_ABL(p) 0
#endif

typedef void (*kernel_fn)(const FastParams);

// plp's second launch (fast_mel.hip, plp_tail_kernel): rows of autocorrelations -> cepstra, in place
struct PlpTailParams {
    float *Y;
    const UttDesc *utts;
    const float *aux;                   // eql[B] | lift[C]
    const float *en;                    // frame energies (use_energy), row = y_off / C + frame
    int B, C, use_energy;
};
hipError_t launch_plp_tail(const PlpTailParams &q, uint32_t nutt, uint32_t max_frames, hipStream_t s);
// the instantiation that serves a plan: nz = rows of samples that hold data (9..16), pair = two frames per transform
kernel_fn fast_spectrogram_kernel_for(int nz, bool dither, bool pcm, bool pair, bool wide, bool xwide);   // fast_spectrogram.hip
kernel_fn fast_mel_kernel_for(int nz, int kind, bool dither, bool pair, bool wide, bool xwide);   // fast_mel.hip
kernel_fn fast_mel_xwide_kernel_for(int nz, int kind, bool dither);                        // fast_mel_xwide.hip

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}

// sum over the 16 lanes of a DPP row; every lane of the row ends with the same bits
__device__ __forceinline__ float row16_sum(float v)
{
    v += dpp_mov<0xB1>(v);              // quad_perm [1,0,3,2]
    v += dpp_mov<0x4E>(v);              // quad_perm [2,3,0,1]
    v += dpp_mov<0x141>(v);             // row_half_mirror
    v += dpp_mov<0x140>(v);             // row_mirror
    return v;
}

// LDS traffic between the lanes of ONE wave needs no s_barrier (a wave's DS operations execute in order);
// this only stops the compiler from moving LDS accesses across the hand-over point.
__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ---- packed f32 arithmetic ------------------------------------------------------------------------------------
// A wave64 f32 VALU instruction occupies its SIMD for 4 cycles whether it is v_fma_f32 or v_pk_fma_f32 (measured:
// tools/ubench/valu_rate.hip, 70 vs 130 TFLOP/s), so complex numbers live in even-aligned VGPR pairs (re, im) and
// every butterfly add, twiddle multiply and front-end step is one packed instruction.  hipcc does not fold the
// (re, im) swaps and sign flips of complex arithmetic into the op_sel / neg modifiers of v_pk_*_f32, so those forms
// are single-instruction asm statements (register operands only: no waits or hazards to manage).
typedef float f2 __attribute__((ext_vector_type(2)));

// a - i b = (a.x + b.y, a.y - b.x)
__device__ __forceinline__ f2 add_mi(f2 a, f2 b)
{
    f2 d;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b));
    return d;
}

// a + i b = (a.x - b.y, a.y + b.x)
__device__ __forceinline__ f2 add_pi(f2 a, f2 b)
{
    f2 d;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(d) : "v"(a), "v"(b));
    return d;
}

// a + conj(b) = (a.x + b.x, a.y - b.y)
__device__ __forceinline__ f2 add_conj(f2 a, f2 b)
{
    f2 d;
    asm("v_pk_add_f32 %0, %1, %2 neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b));
    return d;
}

// -i (a - conj(b)) = (a.y + b.y, b.x - a.x)
__device__ __forceinline__ f2 mi_sub_conj(f2 a, f2 b)
{
    f2 d;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[0,0] neg_hi:[1,0]" : "=v"(d) : "v"(a), "v"(b));
    return d;
}

// (a.x + b.x, a.x - b.x) and (a.y + b.y, a.y - b.y): the sum and the difference of two complex numbers, component by component,
// in one instruction each -- the real parts of (a + b, a - b), then their imaginary parts
__device__ __forceinline__ f2 addsub_lo(f2 a, f2 b)
{
    f2 d;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,0] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b));
    return d;
}

__device__ __forceinline__ f2 addsub_hi(f2 a, f2 b)
{
    f2 d;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,1] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b));
    return d;
}

// complex product z w = (z.x w.x - z.y w.y, z.y w.x + z.x w.y)
__device__ __forceinline__ f2 cmul(f2 z, f2 w)
{
    f2 t, d;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(t) : "v"(z), "v"(w));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[0,1,0]" : "=v"(d) : "v"(z), "v"(w), "v"(t));
    return d;
}

// (a.x, b.x) and (a.y, b.y): builds (re, im) pairs from a pair of reals and a pair of imaginaries in one instruction each
__device__ __forceinline__ f2 pair_lo(f2 a, f2 b)
{
    f2 d;
    asm("v_pk_mov_b32 %0, %1, %2 op_sel:[0,0]" : "=v"(d) : "v"(a), "v"(b));
    return d;
}

__device__ __forceinline__ f2 pair_hi(f2 a, f2 b)
{
    f2 d;
    asm("v_pk_mov_b32 %0, %1, %2 op_sel:[1,1]" : "=v"(d) : "v"(a), "v"(b));
    return d;
}

// The radix-2 joins of the 1024- and 2048-point plans pair lane l with lane l ^ 16 (or l ^ 32): the lower lane of a pair
// computes from both lanes' `a`, the upper one from both lanes' `b`.  v_permlane16_swap_b32 / v_permlane32_swap_b32 (gfx950;
// semantics printed by tools/ubench/permlane_swap.hip) exchange exactly that in one VALU instruction per register:
// afterwards u = (own a | the lower partner's b), v = (the upper partner's a | own b) -- no LDS crossbar trip, no wait, and
// none of the six selects per complex pair that the ds_bpermute form needed to pick what to send and what to keep.
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
template <int ROWS32>
__device__ __forceinline__ void join_swap(const f2 a, const f2 b, f2 &u, f2 &v)
{
    u32x2 rx, ry;
    if (ROWS32)
    {
        rx = __builtin_amdgcn_permlane32_swap(__float_as_uint(a.x), __float_as_uint(b.x), false, false);
        ry = __builtin_amdgcn_permlane32_swap(__float_as_uint(a.y), __float_as_uint(b.y), false, false);
    }
    else
    {
        rx = __builtin_amdgcn_permlane16_swap(__float_as_uint(a.x), __float_as_uint(b.x), false, false);
        ry = __builtin_amdgcn_permlane16_swap(__float_as_uint(a.y), __float_as_uint(b.y), false, false);
    }
    u = f2{__uint_as_float(rx.x), __uint_as_float(ry.x)};
    v = f2{__uint_as_float(rx.y), __uint_as_float(ry.y)};
}

// forward 4-point DFT in place: (a,b,c,d) = x0..x3 -> X0..X3
__device__ __forceinline__ void fft4(f2 &a, f2 &b, f2 &c, f2 &d)
{
    const f2 t0 = a + c, t1 = a - c, t2 = b + d, t3 = b - d;
    a = t0 + t2; c = t0 - t2;
    b = add_mi(t1, t3); d = add_pi(t1, t3);
}

// position of output X[k] after fft16()
__host__ __device__ constexpr int P16(int k) { return (k >> 2) + 4 * (k & 3); }

// forward 16-point DFT on registers, input natural order, output X[k] at position P16(k)
__device__ __forceinline__ void fft16(f2 (&z)[16])
{
    constexpr float C1 = 0.92387953251128675613f, S1 = 0.38268343236508977173f, H = 0.70710678118654752440f;
#pragma unroll
    for (int a = 0; a < 4; ++a) { fft4(z[a], z[a + 4], z[a + 8], z[a + 12]); }
    // position a + 4 kb  *=  W16^(a kb)
    const f2 w1 = {C1, -S1}, w2 = {H, -H}, w3 = {S1, -C1}, w6 = {-H, -H}, w9 = {-C1, S1};
    z[1 + 4] = cmul(z[1 + 4], w1); z[1 + 8] = cmul(z[1 + 8], w2); z[1 + 12] = cmul(z[1 + 12], w3);
    z[2 + 4] = cmul(z[2 + 4], w2); z[2 + 8] = f2{z[2 + 8].y, -z[2 + 8].x}; z[2 + 12] = cmul(z[2 + 12], w6);
    z[3 + 4] = cmul(z[3 + 4], w3); z[3 + 8] = cmul(z[3 + 8], w6); z[3 + 12] = cmul(z[3 + 12], w9);
#pragma unroll
    for (int b = 0; b < 4; ++b) { fft4(z[4 * b], z[4 * b + 1], z[4 * b + 2], z[4 * b + 3]); }
}

// natural log of a positive normal float: exponent and mantissa split so that v_log_f32 (1 ulp on log2) only sees
// [0.5, 1): absolute error ~5e-8 plus the final rounding, instead of 1 ulp of log2(x) ~ 2e-6 at x ~ 1e9
__device__ __forceinline__ float pos_log(float v)
{
    const float m = __builtin_amdgcn_frexp_mantf(v);
    const float e = (float)__builtin_amdgcn_frexp_expf(v);
    return fmaf(e, 0.69314718055994530942f, __log2f(m) * 0.69314718055994530942f);
}

// 1 / x to 1 ulp: v_rcp_f32 and one Newton step (3 instructions; an IEEE division is 10)
__device__ __forceinline__ float fast_rcp(float x)
{
    const float r = __builtin_amdgcn_rcpf(x);
    return fmaf(fmaf(-x, r, 1.0f), r, r);
}

// max(v, FLT_EPSILON) as the one v_max_f32 it is: fmaxf() on a value that comes from LDS is preceded by a canonicalising
// v_max_f32 v, v, v (the compiler cannot prove the bits are a quiet number)
__device__ __forceinline__ float floor_log(float v)
{
    float m;
    asm("v_max_f32 %0, %1, %2" : "=v"(m) : "v"(v), "v"(FLT_EPSILON));
    return pos_log(m);
}

// x^c for x >= 0, 0 < c <= 1 (plp's spectral compression, powf in the reference: c/kaldi_plp.c:355).  exp2(c log2 x) on the
// whole value carries v_log_f32's 1 ulp of log2 x ~ 2e-6 at x ~ 1e9 into the exponent: 4-5 ulp on the result, which
// Levinson-Durbin then amplifies (round 2: HIP 2 x the CPU arithmetic's noise on short plp cases).  Split instead: x = m 2^e,
// log2 m from the hardware on [0.5, 1) (abs error ~1e-7), c e as an exact two-term product (hi + lo by fma), the integer part
// of the exponent applied by ldexp: ~1.5 ulp for 8 more instructions per mel value.
__device__ __forceinline__ float pow_split(float x, float c)
{
    const float m = __builtin_amdgcn_frexp_mantf(x);
    const float e = (float)__builtin_amdgcn_frexp_expf(x);
    const float hi = c * e;
    const float lo = fmaf(c, e, -hi);
    const float ti = __builtin_rintf(hi);
    const float r = (hi - ti) + fmaf(c, __log2f(m), lo);
    return __builtin_amdgcn_ldexpf(__builtin_amdgcn_exp2f(r), (int)ti);
}
#ifndef PLP_IEEE_DIV
#define PLP_IEEE_DIV 1
#endif
#ifndef PLP_POW_SPLIT
#define PLP_POW_SPLIT 1                 // 0: exp2(c log2 x) on the whole value (rounds 1-2), for A/B timing and noise measurements
#endif

// spectrogram: log(max(q / 4, FLT_EPSILON)) for q = 4 |X|^2, exponent and mantissa split as in pos_log: v_log_f32 on the
// whole value is 1 ulp of log2(q) ~ 2e-6 at q ~ 1e9, which measured as 1.2-1.4 x the float32 CPU arithmetic's rms distance
// from float64 (profiles/r2_parity_table.md, round 2); split, the hardware log only sees [0.5, 1) and the one rounding that
// remains is the result's own.  ln(q / 4) = e ln2 + (log2(m) - 2) ln2 with q = m 2^e.
__device__ __forceinline__ float spec_log(float q)
{
    const float v = fmaxf(q, 4.0f * FLT_EPSILON);
    const float m = __builtin_amdgcn_frexp_mantf(v);
    const float e = (float)__builtin_amdgcn_frexp_expf(v);
    return fmaf(e, 0.69314718055994530942f, fmaf(__log2f(m), 0.69314718055994530942f, -1.38629436111989061883f));
}

// two bins at a time: both scale-and-shift steps are v_pk_fma_f32 (12 instructions per pair)
__device__ __forceinline__ f2 spec_log2(float qa, float qb)
{
    const float va = fmaxf(qa, 4.0f * FLT_EPSILON), vb = fmaxf(qb, 4.0f * FLT_EPSILON);
    const f2 l = {__log2f(__builtin_amdgcn_frexp_mantf(va)), __log2f(__builtin_amdgcn_frexp_mantf(vb))};
    const f2 e = {(float)__builtin_amdgcn_frexp_expf(va), (float)__builtin_amdgcn_frexp_expf(vb)};
    const f2 ln2 = {0.69314718055994530942f, 0.69314718055994530942f};
    const f2 t = __builtin_elementwise_fma(l, ln2, f2{-1.38629436111989061883f, -1.38629436111989061883f});
    return __builtin_elementwise_fma(e, ln2, t);
}

// Load the wave's audio slice: `need` samples starting at utterance-relative position s0 (may be negative or run past N
// when snip_edges is off: reflected as c/kaldi_mfcc.c:228).  All arguments but `lane` are wave-uniform.  xu16 non-null:
// the utterance is 16-bit PCM at xu16 and is converted here (exact), otherwise float at xu.
// Interior quads use raw buffer loads: one sample per lane per instruction, lane offset in one VGPR, the step in an
// SGPR, and the descriptor's range check instead of a per-lane bound test.
template <bool PADX = false>
__device__ __forceinline__ void load_span(float *s_aud, const float *xu, const short *xu16, long long s0, int need, long long N, int lane, const int dma_floats = 0)
{
    constexpr int ROW = PADX ? 68 : 64;                                     // floats between the rows of 64 samples (padx)
    const int lofs = PADX ? lane + 2 * (lane >> 5) : lane;
    if (!PADX && !xu16 && ((need + 255) & ~255) <= dma_floats && s0 >= 0 && s0 + need <= N)
    {
        // float audio straight into the slice: buffer_load_dwordx4 ... lds (gfx950) moves 1 KB per wave-instruction from
        // memory into LDS without passing through registers -- 4 instructions for a quad's 880 samples instead of 14 loads
        // and 14 ds_write_b32.  dma_floats = the slice's room in whole instructions (a multiple of 256 floats; a span that needs more takes the
        // register path below); samples past `need` arrive as 0.
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)(xu + s0), 0, need * 4, BUF_FLAGS);
        for (int c = 0; 256 * c < need; ++c)
        {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void *)(s_aud + 256 * c), 16, lane * 16, 1024 * c, 0, 0);
        }
        return;
    }
    if (s0 >= 0 && s0 + need <= N)
    {
        const int nit = (need + 63) >> 6;
        // Sixteen rows of 64 samples are requested back to back and only then parked in LDS, so a quad's audio costs one
        // memory latency, not one per row (a loop of load - wait - store pairs measured 7 latencies for the usual 14 rows).
        // No per-lane bound: the slice is a whole number of 64-float rows, loads past `need` return 0 (range check).
        if (xu16)
        {
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)(xu16 + s0), 0, need * 2, BUF_FLAGS);
            for (int c = 0; c < nit; c += 16)
            {
                unsigned short v[16];
#pragma unroll
                for (int j = 0; j < 16; ++j) { v[j] = __builtin_amdgcn_raw_buffer_load_b16(rs, lane * 2, (c + j) * 128, 0); }
#pragma unroll
                for (int j = 0; j < 16; ++j) { if (c + j < nit) { s_aud[lofs + ROW * (c + j)] = (float)(short)v[j]; } }
            }
        }
        else
        {
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)(xu + s0), 0, need * 4, BUF_FLAGS);
            for (int c = 0; c < nit; c += 16)
            {
                float v[16];
#pragma unroll
                for (int j = 0; j < 16; ++j) { v[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, lane * 4, (c + j) * 256, 0)); }
#pragma unroll
                for (int j = 0; j < 16; ++j) { if (c + j < nit) { s_aud[lofs + ROW * (c + j)] = v[j]; } }
            }
        }
    }
    else
    {
        for (int i = lane; i < need; i += 64)
        {
            long long n = s0 + i;
            while (n < 0 || n >= N) { n = (n < 0) ? -n - 1 : 2 * N - 1 - n; }
            s_aud[PADX ? padx(i) : i] = xu16 ? (float)xu16[n] : xu[n];
        }
    }
}

// Store `nout` floats of the wave's LDS staging to y (contiguous), 256 B per wave-instruction.
__device__ __forceinline__ void store_run(float *y, const float *s_src, int nout, int lane)
{
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)y, 0, nout * 4, BUF_FLAGS);
    const int nit = (nout + 63) >> 6;
    for (int j = 0; j < nit; ++j)
    {
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, s_src[lane + 64 * j]), rs, lane * 4, j * 256, 0);
    }
}

// One frame per 16-lane row: front end + 512-point real FFT + power.  NZ = ceil(L / 32): rows n2 < NZ-1 are full,
// row NZ-1 is cut at L, rows >= NZ are the zero padding (never read, folded at compile time).
// `fa` (8-byte aligned) is the frame's first sample in the wave's LDS slice; `s_tr` is that same slice, reused for the
// transposition once the samples are in registers.
// On return lane k2 (= lane & 15) holds, for k1 = 0..7 (and k1 = 8, meaningful in lane k2 == 0 only):
//   pa[k1] = 4 |X[k]|^2, pb[k1] = 4 |X[256 - k]|^2 with k = k2 + 16 k1 (the factor 4 is folded into what consumes them);
//   rawe = the frame's (floored) energy.
// 256-point complex FFT of z[n], n = n1 + 16 n2, lane n1 holding z[n1 + 16 n2] in z[n2]: 16-point FFT over n2 in registers,
// twiddle W256^(n1 k2), transposition through the wave's LDS slice, 16-point FFT over n1.  On return lane k2 (= n1) holds
// Z[k2 + 16 k1] at position P16(k1).
template <bool HALF, int TW1 = TAB_TW1>
__device__ __forceinline__ void fft256(f2 (&z)[16], const float *s_tab, float *s_tr, const int lane)
{
    const int f = lane >> 4, n1 = lane & 15;
    fft16(z);
    const f2 *tw1 = reinterpret_cast<const f2 *>(s_tab + TW1);
#pragma unroll
    for (int k2 = 1; k2 < 16; ++k2) { z[P16(k2)] = cmul(z[P16(k2)], tw1[k2 * 16 + n1]); }
    wave_sync();                                                            // every lane has its samples: the slice becomes the transposition image
    const int k2 = n1;                                                      // the lane's role after the transpose
    if (HALF)
    {
        // real parts then imaginary parts (a 1 KB image per frame keeps 7 workgroups on a CU): element (k2, n) lives at
        // 16 (k2 ^ (f & 1)) + 4 ((n>>2) ^ (k2>>2)) + (n&3), which makes the b32 writes and the b128 reads conflict-free: the
        // swizzle spreads a frame's 16 lanes over 16 banks, and the row flip of the odd frames puts them on the other 16
        // (a b32 access is served 32 lanes = two frames at a time; 256 floats per image alone left both on the same banks:
        // 64 conflict cycles per quad, measured with tools/lds_by_phase.sh)
        float *trf = s_tr + f * 256;
        const int fx = (f & 1) << 4;
        float *trf_e = trf + fx, *trf_o = trf - fx;                         // rows kk ^ 1 of the odd frames: + 16 floats for even kk, - 16 for odd kk
        int woff[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) { woff[c] = (((n1 >> 2) ^ c) << 2) + (n1 & 3); }
        const float *rrow = trf + ((k2 * 16) ^ fx);
        const int rq = k2 >> 2;
        float4 rr[4];
#pragma unroll
        for (int c = 0; c < 4; ++c)                                          // rows 4c, 4c+2 | 4c+1, 4c+3: each pair one ds_write2_b32
        {
            trf_e[(4 * c) * 16 + woff[c]] = z[P16(4 * c)].x; trf_e[(4 * c + 2) * 16 + woff[c]] = z[P16(4 * c + 2)].x;
            trf_o[(4 * c + 1) * 16 + woff[c]] = z[P16(4 * c + 1)].x; trf_o[(4 * c + 3) * 16 + woff[c]] = z[P16(4 * c + 3)].x;
        }
        wave_sync();
#pragma unroll
        for (int q = 0; q < 4; ++q) { rr[q] = *reinterpret_cast<const float4 *>(rrow + ((q ^ rq) << 2)); }
        wave_sync();
#pragma unroll
        for (int c = 0; c < 4; ++c)                                          // rows 4c, 4c+2 | 4c+1, 4c+3: each pair one ds_write2_b32
        {
            trf_e[(4 * c) * 16 + woff[c]] = z[P16(4 * c)].y; trf_e[(4 * c + 2) * 16 + woff[c]] = z[P16(4 * c + 2)].y;
            trf_o[(4 * c + 1) * 16 + woff[c]] = z[P16(4 * c + 1)].y; trf_o[(4 * c + 3) * 16 + woff[c]] = z[P16(4 * c + 3)].y;
        }
        wave_sync();
#pragma unroll
        for (int q = 0; q < 4; ++q)
        {
            const float4 v = *reinterpret_cast<const float4 *>(rrow + ((q ^ rq) << 2));
            const f2 r01 = {rr[q].x, rr[q].y}, r23 = {rr[q].z, rr[q].w}, i01 = {v.x, v.y}, i23 = {v.z, v.w};
            z[4 * q] = pair_lo(r01, i01); z[4 * q + 1] = pair_hi(r01, i01); z[4 * q + 2] = pair_lo(r23, i23); z[4 * q + 3] = pair_hi(r23, i23);
        }
    }
    else
    {
        // complex image (2 KB per frame): element (k2, n) lives at float offset 32 k2 + 4 ((n>>1) ^ (k2>>1)) + 2 (n&1):
        // b64 writes and b128 reads are both conflict-free and the pairs come back as pairs
        float *trf = s_tr + f * 512;
        const int wlo = 2 * (n1 & 1), wm = n1 >> 1;
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) { *reinterpret_cast<f2 *>(trf + 32 * kk + 4 * (wm ^ (kk >> 1)) + wlo) = z[P16(kk)]; }
        wave_sync();
        const float *rrow = trf + 32 * k2;
        const int rg = k2 >> 1;
#pragma unroll
        for (int m = 0; m < 8; ++m)
        {
            const float4 v = *reinterpret_cast<const float4 *>(rrow + 4 * (m ^ rg));
            z[2 * m] = f2{v.x, v.y}; z[2 * m + 1] = f2{v.z, v.w};
        }
    }
    wave_sync();
    fft16(z);                                                               // Z[k2 + 16 k1] at position P16(k1)

}

template <int NZ, bool HALF, bool DITHER>
__device__ __forceinline__ void frame_power(const float *fa, const float *s_tab, float *s_tr, const int lane, const FastParams &p,
                                            const unsigned long long dkey, float (&pa)[9], float (&pb)[9], float &rawe)
{
    const int n1 = lane & 15;
    const int L = p.L;
    const int llast = 2 * n1 + 32 * (NZ - 1);
    const bool ve = llast < L, vo = llast + 1 < L;                          // validity of the lane's two samples in row NZ-1
    f2 z[16];
    f2 sum2 = {0.0f, 0.0f};
#pragma unroll
    for (int n2 = 0; n2 < 16; ++n2)
    {
        if (n2 < NZ)
        {
            f2 v = *reinterpret_cast<const f2 *>(fa + 2 * n1 + 32 * n2);
            if (n2 == NZ - 1) { v.x = ve ? v.x : 0.0f; v.y = vo ? v.y : 0.0f; }
            z[n2] = v;
            sum2 += v;
        }
        else { z[n2] = f2{0.0f, 0.0f}; }
    }
    if (DITHER)                                                             // c/kaldi_mfcc.c:243-262; the lane's pairs are i = n1 + 16 n2
    {
        sum2 = f2{0.0f, 0.0f};
#pragma unroll
        for (int n2 = 0; n2 < NZ; ++n2)
        {
            float g0, g1;
            dither_pair(dither_bits(dkey, (unsigned)(n1 + 16 * n2)), p.dither, g0, g1);
            z[n2] += f2{g0, g1};
            if (n2 == NZ - 1) { z[n2].x = ve ? z[n2].x : 0.0f; z[n2].y = vo ? z[n2].y : 0.0f; }
            sum2 += z[n2];
        }
    }
    float mean = 0.0f;
    if (p.dc0) { mean = row16_sum(sum2.x + sum2.y) * p.invL; }             // c/kaldi_mfcc.c:265-271
    const f2 mean2 = {mean, mean};
    const float npre = -p.preemph;
    const f2 *win = reinterpret_cast<const f2 *>(s_tab + TAB_WIN);
    const float *fprev = fa + ((n1 > 0) ? 2 * n1 - 1 : 0);                  // sample before the lane's even sample of row 0 (itself for l = 0)
    f2 e2 = {0.0f, 0.0f};
#pragma unroll
    for (int n2 = 0; n2 < NZ; ++n2)
    {
        z[n2] -= mean2;
        if (n2 == NZ - 1) { z[n2].x = ve ? z[n2].x : 0.0f; z[n2].y = vo ? z[n2].y : 0.0f; }
    }
    if (p.want_pre)                                                         // c/kaldi_mfcc.c:274-279 (a wave-uniform branch, not a select per row)
    {
#pragma unroll
        for (int n2 = 0; n2 < NZ; ++n2) { e2 = __builtin_elementwise_fma(z[n2], z[n2], e2); }
    }
    if (DITHER)
    {
        // the dithered samples exist in registers only: the sample before a lane's even one is the odd one of the lane
        // to its left in the same row (DPP rotate within the frame's 16 lanes), for lane 0 that of lane 15 one row up
        float up = z[0].x;                                                  // l = 0: y[0] = d[0] - p d[0]
#pragma unroll
        for (int n2 = 0; n2 < NZ; ++n2)
        {
            const float left = dpp_mov<0x121>(z[n2].y);                     // row_ror:1
            const float xpm = (n1 == 0) ? up : left;
            up = left;
            f2 y;
            y.x = fmaf(npre, xpm, z[n2].x);
            y.y = fmaf(npre, z[n2].x, z[n2].y);
            y *= win[n1 + 16 * n2];
            if (n2 == NZ - 1) { y.x = ve ? y.x : 0.0f; y.y = vo ? y.y : 0.0f; }
            z[n2] = y;
        }
    }
    else
#pragma unroll
    for (int n2 = 0; n2 < NZ; ++n2)
    {
        // pre-emphasis: y[l] = d[l] - p d[l-1], y[0] = d[0] - p d[0]         c/kaldi_mfcc.c:282-288
        const float xpm = ((n2 == 0) ? fprev[0] : fa[2 * n1 + 32 * n2 - 1]) - mean;
        f2 y;
        y.x = fmaf(npre, xpm, z[n2].x);
        y.y = fmaf(npre, z[n2].x, z[n2].y);
        y *= win[n1 + 16 * n2];                                             // window: c/kaldi_mfcc.c:291 (table is zero beyond L)
        if (n2 == NZ - 1) { y.x = ve ? y.x : 0.0f; y.y = vo ? y.y : 0.0f; } // the slice beyond the quad's last sample is not initialised
        z[n2] = y;
    }
    if (p.want_post)                                                        // c/kaldi_mfcc.c:295-301
    {
#pragma unroll
        for (int n2 = 0; n2 < NZ; ++n2) { e2 = __builtin_elementwise_fma(z[n2], z[n2], e2); }
    }
    rawe = fmaxf(row16_sum(e2.x + e2.y), FLT_EPSILON);

    fft256<HALF>(z, s_tab, s_tr, lane);
    const int k2 = n1;                                                      // the lane's role after the transpose

    // untangle: partner lane (16 - k2) mod 16 of the same row holds Z[256 - k] in register 15 - k1; lane 0 is its own
    // partner with register (16 - k1) mod 16
    const int paddr = ((lane & 48) | ((16 - k2) & 15)) << 2;
    const bool lane0 = (k2 == 0);
    const f2 *tw2 = reinterpret_cast<const f2 *>(s_tab + TAB_TW2);
    {
        // bin 128 (lane 0): Z[128] pairs with itself and exp(-i pi 128/256) = -i, so X[128] = conj Z[128]
        const f2 q8 = z[P16(8)] * z[P16(8)];
        pa[8] = 4.0f * (q8.x + q8.y);
        pb[8] = pa[8];
    }
#pragma unroll
    for (int k1 = 0; k1 < 8; ++k1)
    {
        const f2 zo = z[P16((15 - k1) & 15)], zs = z[P16((16 - k1) & 15)];
        f2 zp;
        zp.x = __int_as_float(__builtin_amdgcn_ds_bpermute(paddr, __float_as_int(zo.x)));
        zp.y = __int_as_float(__builtin_amdgcn_ds_bpermute(paddr, __float_as_int(zo.y)));
        zp.x = lane0 ? zs.x : zp.x;
        zp.y = lane0 ? zs.y : zp.y;
        const f2 zk = z[P16(k1)];
        const f2 e = add_conj(zk, zp);                                      // 2E = Z[k] + conj Z[256-k]
        const f2 t = cmul(mi_sub_conj(zk, zp), tw2[k1 * 16 + k2]);          // 2O exp(-i pi k / 256), 2O = -i (Z[k] - conj Z[256-k])
        // X[k] = e + t and X[256-k] = conj(e - t): their real parts as one packed pair, their imaginary parts as another, so that
        // both powers come out of one v_pk_mul and one v_pk_fma (round 3: 4 instructions where a = e + t, b = e - t, a a, b b and
        // two adds took 6; the second square is fused, |X|^2 = fma(im, im, re re))
        const f2 re2 = addsub_lo(e, t), im2 = addsub_hi(e, t);
        const f2 q = __builtin_elementwise_fma(im2, im2, re2 * re2);
        pa[k1] = q.x;
        pb[k1] = q.y;
    }
}

// 1024-point plans: ONE real frame of 513 .. 1024 samples on 32 lanes.  Its even samples xe[m] = x[2m] and its odd samples
// xo[m] = x[2m+1] are two real sequences of 512 points: lanes 16 g .. 16 g + 15 with g even take the even ones, g odd the odd
// ones, each exactly as frame_power treats a 512-point frame (packing, 256-point complex FFT, untangle -- but keeping the
// complex bins), and
//   X[k] = E[k] + W1024^k O[k],     X[512-k] = conj(E[k] - W1024^k O[k]),     k = 0 .. 256
// joins the halves after ONE exchange between lanes l and l ^ 16: the untangle leaves a = 2 S[k] and b = 2 conj S[256-k]
// (S = E or O) in every lane; the even group computes bins k and 512-k from (aE, aO), the odd group bins 256-k and 256+k
// from (bE, bO) with conj W1024^(256-k) = i W1024^k.
// NZ = ceil(L / 64) rows of 64 samples hold data; fa = the frame's first sample in the slice.  With k = k2 + 16 k1, k1 < 8:
//   even-group lanes: pa[k1] = 4 |X[k]|^2,     pb[k1] = 4 |X[512-k]|^2;  lane k2 = 0 also pa[8], pb[8] = bins 128, 384
//   odd-group lanes:  pa[k1] = 4 |X[256-k]|^2, pb[k1] = 4 |X[256+k]|^2
template <int NZ, bool HALF, bool DITHER>
__device__ __forceinline__ void frame_power_wide(const float *fa, const float *s_tab, float *s_tr, const int lane, const FastParams &p,
                                                 const unsigned long long dkey, float (&pa)[9], float (&pb)[9], float &rawe)
{
    const int n1 = lane & 15, h = (lane >> 4) & 1;
    const int L = p.L;
    const int i00 = 4 * n1 + h;                                             // the lane's samples of row n2: i00 + 64 n2 and that + 2
    const int ilast = i00 + 64 * (NZ - 1);
    const bool v0 = ilast < L, v1 = ilast + 2 < L;
    const float *win = s_tab + TABW_WIN;
    // the lane's two samples of a row travel as one packed pair: sums, mean, pre-emphasis and window are v_pk_* instructions
    f2 z[16];
    f2 sum2 = {0.0f, 0.0f};
#pragma unroll
    for (int n2 = 0; n2 < 16; ++n2)
    {
        if (n2 < NZ)
        {
            f2 v = {fa[i00 + 64 * n2], fa[i00 + 64 * n2 + 2]};
            if (n2 == NZ - 1) { v.x = v0 ? v.x : 0.0f; v.y = v1 ? v.y : 0.0f; }
            z[n2] = v;
            sum2 += v;
        }
        else { z[n2] = f2{0.0f, 0.0f}; }
    }
    if (DITHER)                                                             // c/kaldi_mfcc.c:243-262
    {
        // samples i00 + 64 n2 and that + 2 are the members of parity h of the Box-Muller pairs 2 n1 + 32 n2 and the next one
        // (the lane of the other group draws the same pairs and takes the other member)
        sum2 = f2{0.0f, 0.0f};
#pragma unroll
        for (int n2 = 0; n2 < NZ; ++n2)
        {
            const unsigned pr = (unsigned)(2 * n1 + 32 * n2);
            z[n2].x += dither_one(dither_bits(dkey, pr), p.dither, h);
            z[n2].y += dither_one(dither_bits(dkey, pr + 1u), p.dither, h);
            if (n2 == NZ - 1) { z[n2].x = v0 ? z[n2].x : 0.0f; z[n2].y = v1 ? z[n2].y : 0.0f; }
            sum2 += z[n2];
        }
    }
    float mean = 0.0f;
    if (p.dc0)                                                              // c/kaldi_mfcc.c:265-271
    {
        float tot = row16_sum(sum2.x + sum2.y);
        tot += __shfl_xor(tot, 16, 64);
        mean = tot * p.invL;
    }
    const float npre = -p.preemph;
    const f2 mean2 = {mean, mean}, npre2 = {npre, npre};
    f2 e2 = {0.0f, 0.0f};
#pragma unroll
    for (int n2 = 0; n2 < NZ; ++n2)
    {
        z[n2] -= mean2;
        if (n2 == NZ - 1) { z[n2].x = v0 ? z[n2].x : 0.0f; z[n2].y = v1 ? z[n2].y : 0.0f; }
    }
    if (p.want_pre)                                                         // c/kaldi_mfcc.c:274-279
    {
#pragma unroll
        for (int n2 = 0; n2 < NZ; ++n2) { e2 = __builtin_elementwise_fma(z[n2], z[n2], e2); }
    }
    if (DITHER)
    {
        // the dithered samples exist in registers only.  The sample before the lane's first one is the second one of the odd
        // group's lane n1 - 1 (even group; lane 0: that of lane 15 one row up) or the first one of the even group's lane n1
        // (odd group); the sample before its second one is the first (even group) / second (odd group) one of the other
        // group's lane n1.
        const int xa = (h ? (lane ^ 16) : ((lane & 32) | 16 | ((n1 - 1) & 15))) << 2, xb = (lane ^ 16) << 2;
        float up = z[0].x;                                                  // l = 0: y[0] = d[0] - p d[0]
#pragma unroll
        for (int n2 = 0; n2 < NZ; ++n2)
        {
            // what a lane SENDS: even-group lanes are asked for .x through xa and .y through xb, odd-group lanes the reverse
            const float fa_ = __int_as_float(__builtin_amdgcn_ds_bpermute(xa, __float_as_int(h ? z[n2].y : z[n2].x)));
            const float fb_ = __int_as_float(__builtin_amdgcn_ds_bpermute(xb, __float_as_int(h ? z[n2].x : z[n2].y)));
            const float xp0 = (h == 0 && n1 == 0) ? up : fa_;
            up = fa_;
            const int i0 = i00 + 64 * n2;
            f2 y = __builtin_elementwise_fma(npre2, f2{xp0, fb_}, z[n2]) * f2{win[i0], win[i0 + 2]};
            if (n2 == NZ - 1) { y.x = v0 ? y.x : 0.0f; y.y = v1 ? y.y : 0.0f; }
            z[n2] = y;
        }
    }
    else
#pragma unroll
    for (int n2 = 0; n2 < NZ; ++n2)
    {
        // pre-emphasis y[l] = d[l] - p d[l-1], y[0] = d[0] - p d[0] (c/kaldi_mfcc.c:282-288), window (:291)
        const int i0 = i00 + 64 * n2;
        const f2 xp = f2{fa[(i0 > 0) ? i0 - 1 : 0], fa[i0 + 1]} - mean2;
        f2 y = __builtin_elementwise_fma(npre2, xp, z[n2]) * f2{win[i0], win[i0 + 2]};
        if (n2 == NZ - 1) { y.x = v0 ? y.x : 0.0f; y.y = v1 ? y.y : 0.0f; }
        z[n2] = y;
    }
    if (p.want_post)                                                        // c/kaldi_mfcc.c:295-301
    {
#pragma unroll
        for (int n2 = 0; n2 < NZ; ++n2) { e2 = __builtin_elementwise_fma(z[n2], z[n2], e2); }
    }
    {
        float tot = row16_sum(e2.x + e2.y);
        tot += __shfl_xor(tot, 16, 64);
        rawe = fmaxf(tot, FLT_EPSILON);
    }

    fft256<HALF, TABW_TW1>(z, s_tab, s_tr, lane);
    const int k2 = n1;

    // untangle within the 16-lane group (as frame_power), keeping a = 2 S[k] and b = 2 conj S[256-k]; then the exchange
    const int paddr = ((lane & 48) | ((16 - k2) & 15)) << 2;
    const int xaddr = (lane ^ 16) << 2;
    const bool lane0 = (k2 == 0);
    const f2 *tw2 = reinterpret_cast<const f2 *>(s_tab + TABW_TW2);
    const f2 *tw3 = reinterpret_cast<const f2 *>(s_tab + TABW_TW3);
    {
        // bin 128 of the halves (lane 0): S[128] = conj Z[128]; X[128], X[384] = E[128] +- exp(-i pi / 4) O[128]
        const f2 z8 = z[P16(8)];
        const f2 a8 = {2.0f * z8.x, -2.0f * z8.y};
        f2 o8;
        o8.x = __int_as_float(__builtin_amdgcn_ds_bpermute(xaddr, __float_as_int(a8.x)));
        o8.y = __int_as_float(__builtin_amdgcn_ds_bpermute(xaddr, __float_as_int(a8.y)));
        const f2 w = {0.70710678118654752440f, -0.70710678118654752440f};
        const f2 t = cmul(o8, w);
        const f2 xa = a8 + t, xb = a8 - t;
        const f2 qa = xa * xa, qb = xb * xb;
        pa[8] = qa.x + qa.y;
        pb[8] = qb.x + qb.y;
    }
#pragma unroll
    for (int k1 = 0; k1 < 8; ++k1)
    {
        const f2 zo = z[P16((15 - k1) & 15)], zs = z[P16((16 - k1) & 15)];
        f2 zp;
        zp.x = __int_as_float(__builtin_amdgcn_ds_bpermute(paddr, __float_as_int(zo.x)));
        zp.y = __int_as_float(__builtin_amdgcn_ds_bpermute(paddr, __float_as_int(zo.y)));
        zp.x = lane0 ? zs.x : zp.x;
        zp.y = lane0 ? zs.y : zp.y;
        const f2 zk = z[P16(k1)];
        const f2 ee = add_conj(zk, zp);
        const f2 tt = cmul(mi_sub_conj(zk, zp), tw2[k1 * 16 + k2]);
        const f2 a = ee + tt, b = ee - tt;                                  // 2 S[k], 2 conj S[256-k]
        f2 u, v;
        join_swap<0>(a, b, u, v);
        const f2 w = tw3[k1 * 16 + k2];
        const f2 g = h ? f2{-w.y, w.x} : w;                                 // odd group: conj W1024^(256-k) = i W1024^k
        const f2 t = cmul(v, g);
        const f2 xa = u + t, xb = u - t;
        const f2 qa = xa * xa, qb = xb * xb;
        pa[k1] = qa.x + qa.y;
        pb[k1] = qb.x + qb.y;
    }
}

// 2048-point plans: ONE real frame of 1025 .. 2048 samples on the wave's 64 lanes.  The four decimated sequences
// s_g[m] = x[4 m + g] (g = 0..3, 512 real points each) go to the four 16-lane groups, each exactly as frame_power treats a
// 512-point frame (packing, 256-point complex FFT, untangle keeping the complex bins: a = 2 S_g[k], b = 2 conj S_g[256-k]),
// and two radix-2 joins follow, each ONE exchange of one complex number per lane and k:
//   step 1, lanes l and l ^ 32 (as frame_power_wide):  E1 = FFT1024 of the even samples = S_0 + W1024^k S_2,  O1 = that of
//     the odd samples = S_1 + W1024^k S_3;  groups 0, 1 end with  xa = 2 F[k],  xb = 2 conj F[512-k];  groups 2, 3 with
//     xa = 2 conj F[256-k],  xb = 2 F[256+k]   (F = E1 in groups 0, 2; O1 in groups 1, 3);
//   step 2, lanes l and l ^ 16:  X[j] = E1[j] + W2048^j O1[j],  X[1024-j] = conj(E1[j] - W2048^j O1[j])  with j = k (group 0),
//     512-k (group 1: conjugated operands, conj W2048^(512-k) = i W2048^k), 256-k (group 2: exp(i pi/4) W2048^k) and 256+k
//     (group 3: exp(-i pi/4) W2048^k).
// NZ = ceil(L / 128) rows of 128 samples hold data; fa = the frame's first sample in the slice.  With k = k2 + 16 k1 (k1 < 8;
// k1 = 8: k = 128, meaningful in lanes k2 = 0 of groups 0 and 1 only) the lanes return 4 |X|^2 of the bins
//   group 0: pa = k,       pb = 1024-k        group 1: pa = 512-k,   pb = 512+k
//   group 2: pa = 256-k,   pb = 768+k         group 3: pa = 256+k,   pb = 768-k
// (k = 0: groups 1 and 3 repeat bins 512 and 256 / 768; k = 128: groups 2 and 3 repeat groups 0 and 1.)
// WING: the window factors come from the plan's table in memory (p.tab, L1-resident) instead of from LDS; they are requested
// right after the samples and arrive during the mean reduction.
template <int NZ, bool HALF, bool DITHER, bool WING = false>
__device__ __forceinline__ void frame_power_xwide(const float *fa, const float *s_tab, float *s_tr, const int lane, const FastParams &p,
                                                  const unsigned long long dkey, float (&pa)[9], float (&pb)[9], float &rawe)
{
    const int n1 = lane & 15, g = lane >> 4;
    const int L = p.L;
    const int i00 = 8 * n1 + g;                                             // the lane's samples of row n2: i00 + 128 n2 and that + 4
    const int ilast = i00 + 128 * (NZ - 1);
    const bool v0 = ilast < L, v1 = ilast + 4 < L;
    // padded positions (padx): a row of 128 samples is 136 floats on; i00 .. i00 + 4 share a block of 32, i00 - 1 may not
    const int q00 = padx(i00);
    const int qm0 = (i00 > 0) ? padx(i00 - 1) : 0, qm = (i00 > 0) ? padx(i00 - 1) : -3;      // sample before the first one: row 0 (itself for l = 0), rows n2 > 0
    const float *win = s_tab + TABX_WIN;
    f2 wg[WING ? NZ : 1];
    if (WING)
    {
        const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void *)(p.tab + TABX_WIN), 0, (TABX_FLOATS - TABX_WIN) * 4, BUF_FLAGS);
#pragma unroll
        for (int n2 = 0; n2 < NZ; ++n2)
        {
            wg[n2].x = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rw, q00 * 4, 136 * 4 * n2, 0));
            wg[n2].y = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rw, q00 * 4, 136 * 4 * n2 + 16, 0));
        }
    }
    // the lane's two samples of a row travel as one packed pair: sums, mean, pre-emphasis and window are v_pk_* instructions
    f2 z[16];
    f2 sum2 = {0.0f, 0.0f};
#pragma unroll
    for (int n2 = 0; n2 < 16; ++n2)
    {
        if (n2 < NZ)
        {
            f2 v = {fa[q00 + 136 * n2], fa[q00 + 136 * n2 + 4]};
            if (n2 == NZ - 1) { v.x = v0 ? v.x : 0.0f; v.y = v1 ? v.y : 0.0f; }
            z[n2] = v;
            sum2 += v;
        }
        else { z[n2] = f2{0.0f, 0.0f}; }
    }
    if (DITHER)                                                             // c/kaldi_mfcc.c:243-262
    {
        // samples i00 + 128 n2 and that + 4 are the members of parity g & 1 of the Box-Muller pairs 4 n1 + (g >> 1) + 64 n2 and
        // that + 2 (the lane of the neighbouring group draws the same pairs and takes the other member)
        sum2 = f2{0.0f, 0.0f};
#pragma unroll
        for (int n2 = 0; n2 < NZ; ++n2)
        {
            const unsigned pr = (unsigned)(4 * n1 + (g >> 1) + 64 * n2);
            z[n2].x += dither_one(dither_bits(dkey, pr), p.dither, g & 1);
            z[n2].y += dither_one(dither_bits(dkey, pr + 2u), p.dither, g & 1);
            if (n2 == NZ - 1) { z[n2].x = v0 ? z[n2].x : 0.0f; z[n2].y = v1 ? z[n2].y : 0.0f; }
            sum2 += z[n2];
        }
    }
    float mean = 0.0f;
    if (p.dc0)                                                              // c/kaldi_mfcc.c:265-271
    {
        float tot = row16_sum(sum2.x + sum2.y);
        tot += __shfl_xor(tot, 16, 64);
        tot += __shfl_xor(tot, 32, 64);
        mean = tot * p.invL;
    }
    const float npre = -p.preemph;
    const f2 mean2 = {mean, mean}, npre2 = {npre, npre};
    f2 e2 = {0.0f, 0.0f};
#pragma unroll
    for (int n2 = 0; n2 < NZ; ++n2)
    {
        z[n2] -= mean2;
        if (n2 == NZ - 1) { z[n2].x = v0 ? z[n2].x : 0.0f; z[n2].y = v1 ? z[n2].y : 0.0f; }
    }
    if (p.want_pre)                                                         // c/kaldi_mfcc.c:274-279
    {
#pragma unroll
        for (int n2 = 0; n2 < NZ; ++n2) { e2 = __builtin_elementwise_fma(z[n2], z[n2], e2); }
    }
    if (DITHER)
    {
        // the dithered samples exist in registers only.  The sample before the lane's first one is the first one of group
        // g - 1's lane n1 (g > 0) or the second one of group 3's lane n1 - 1 (g = 0; lane 0: that of lane 15 one row up); the
        // sample before its second one is the second one of group g - 1's lane n1 (g > 0) or the first one of group 3's lane n1
        const int xa = ((g > 0) ? lane - 16 : (48 | ((n1 - 1) & 15))) << 2, xb = ((g > 0) ? lane - 16 : (48 | n1)) << 2;
        float up = z[0].x;                                                  // l = 0: y[0] = d[0] - p d[0]
#pragma unroll
        for (int n2 = 0; n2 < NZ; ++n2)
        {
            // what a lane SENDS: groups 0..2 are asked for .x through xa and .y through xb, group 3 the reverse
            const float fa_ = __int_as_float(__builtin_amdgcn_ds_bpermute(xa, __float_as_int((g == 3) ? z[n2].y : z[n2].x)));
            const float fb_ = __int_as_float(__builtin_amdgcn_ds_bpermute(xb, __float_as_int((g == 3) ? z[n2].x : z[n2].y)));
            const float xp0 = (g == 0 && n1 == 0) ? up : fa_;
            up = fa_;
            const int q0 = q00 + 136 * n2;
            f2 y = __builtin_elementwise_fma(npre2, f2{xp0, fb_}, z[n2]) * (WING ? wg[n2] : f2{win[q0], win[q0 + 4]});
            if (n2 == NZ - 1) { y.x = v0 ? y.x : 0.0f; y.y = v1 ? y.y : 0.0f; }
            z[n2] = y;
        }
    }
    else
#pragma unroll
    for (int n2 = 0; n2 < NZ; ++n2)
    {
        // pre-emphasis y[l] = d[l] - p d[l-1], y[0] = d[0] - p d[0] (c/kaldi_mfcc.c:282-288), window (:291)
        const int q0 = q00 + 136 * n2;
        const f2 xp = f2{fa[(n2 == 0) ? qm0 : qm + 136 * n2], fa[q0 + 3]} - mean2;
        f2 y = __builtin_elementwise_fma(npre2, xp, z[n2]) * (WING ? wg[n2] : f2{win[q0], win[q0 + 4]});
        if (n2 == NZ - 1) { y.x = v0 ? y.x : 0.0f; y.y = v1 ? y.y : 0.0f; }
        z[n2] = y;
    }
    if (p.want_post)                                                        // c/kaldi_mfcc.c:295-301
    {
#pragma unroll
        for (int n2 = 0; n2 < NZ; ++n2) { e2 = __builtin_elementwise_fma(z[n2], z[n2], e2); }
    }
    {
        float tot = row16_sum(e2.x + e2.y);
        tot += __shfl_xor(tot, 16, 64);
        tot += __shfl_xor(tot, 32, 64);
        rawe = fmaxf(tot, FLT_EPSILON);
    }

    fft256<HALF, TABX_TW1>(z, s_tab, s_tr, lane);
    const int k2 = n1;

    // untangle within the 16-lane group (as frame_power), then the two joins
    const int paddr = ((lane & 48) | ((16 - k2) & 15)) << 2;
    const bool lane0 = (k2 == 0);
    const bool h1 = (g & 2) != 0;
    const f2 *tw2 = reinterpret_cast<const f2 *>(s_tab + TABX_TW2);
    const f2 *tw3 = reinterpret_cast<const f2 *>(s_tab + TABX_TW3);
    const f2 *tw4 = reinterpret_cast<const f2 *>(s_tab + TABX_TW4);
    constexpr float H = 0.70710678118654752440f;
    const f2 ph = (g == 0) ? f2{1.0f, 0.0f} : ((g == 1) ? f2{0.0f, 1.0f} : ((g == 2) ? f2{H, H} : f2{H, -H}));
#pragma unroll
    for (int k1 = 0; k1 < 9; ++k1)
    {
        f2 a, b;
        if (k1 < 8)
        {
            const f2 zo = z[P16((15 - k1) & 15)], zs = z[P16((16 - k1) & 15)];
            f2 zp;
            zp.x = __int_as_float(__builtin_amdgcn_ds_bpermute(paddr, __float_as_int(zo.x)));
            zp.y = __int_as_float(__builtin_amdgcn_ds_bpermute(paddr, __float_as_int(zo.y)));
            zp.x = lane0 ? zs.x : zp.x;
            zp.y = lane0 ? zs.y : zp.y;
            const f2 zk = z[P16(k1)];
            const f2 ee = add_conj(zk, zp);
            const f2 tt = cmul(mi_sub_conj(zk, zp), tw2[k1 * 16 + k2]);
            a = ee + tt; b = ee - tt;                                       // 2 S[k], 2 conj S[256-k]
        }
        else
        {
            // bin 128 of the quarters (lane 0): S[128] = conj Z[128], so a = 2 conj Z[128] and b = 2 conj S[128] = 2 Z[128]
            const f2 z8 = z[P16(8)];
            a = f2{2.0f * z8.x, -2.0f * z8.y}; b = f2{2.0f * z8.x, 2.0f * z8.y};
        }
        // step 1 (lanes l ^ 32)
        f2 u1, v1c;
        join_swap<1>(a, b, u1, v1c);
        const f2 w3 = tw3[k1 * 16 + k2];
        const f2 g1 = h1 ? f2{-w3.y, w3.x} : w3;                            // groups 2, 3: conj W1024^(256-k) = i W1024^k
        const f2 t1 = cmul(v1c, g1);
        const f2 xa = u1 + t1, xb = u1 - t1;
        // step 2 (lanes l ^ 16)
        f2 u2, v2c;
        join_swap<0>(xa, xb, u2, v2c);
        const f2 t2 = cmul(v2c, cmul(tw4[k1 * 16 + k2], ph));
        const f2 ya = u2 + t2, yb = u2 - t2;
        const f2 qa = ya * ya, qb = yb * yb;
        pa[k1] = qa.x + qa.y;
        pb[k1] = qb.x + qb.y;
    }
}

// 256-point plans: TWO real frames per complex transform.  z[n] = xA[n] + i xB[n] (n < 256, frames A and B
// of the same utterance, B one frame shift after A) goes through the same 256-point complex FFT, and
//   2 XA[k] = Z[k] + conj Z[256-k],   2 XB[k] = -i (Z[k] - conj Z[256-k])
// are the untangle step without its twiddle: a wave carries 8 frames (4 pairs, 16 lanes per pair) for the arithmetic of 4.
// NZ = ceil(L / 16) rows of 16 samples hold data.  fa = frame A's first sample in the slice, bofs = floats from there to
// frame B's (0 when B does not exist: its lanes then repeat A and their results are dropped).
// pa[k1] / pb[k1] = 4 |XA[k]|^2 / 4 |XB[k]|^2 for k = k2 + 16 k1 (k1 < 8) and, on lane k2 = 0, k1 = 8: bin 128.
template <int NZ, bool HALF, bool DITHER>
__device__ __forceinline__ void frame_power_pair(const float *fa, const int bofs, const float *s_tab, float *s_tr, const int lane, const FastParams &p,
                                                 const unsigned long long dkey_a, const unsigned long long dkey_b,
                                                 float (&pa)[9], float (&pb)[9], f2 &rawe)
{
    const int n1 = lane & 15;
    const int L = p.L;
    const bool vl = n1 + 16 * (NZ - 1) < L;                                 // validity of the lane's sample in row NZ-1
    const float *fb = fa + bofs;
    f2 z[16];
    f2 sum2 = {0.0f, 0.0f};
#pragma unroll
    for (int n2 = 0; n2 < 16; ++n2)
    {
        if (n2 < NZ)
        {
            f2 v = {fa[n1 + 16 * n2], fb[n1 + 16 * n2]};
            if (n2 == NZ - 1) { v.x = vl ? v.x : 0.0f; v.y = vl ? v.y : 0.0f; }
            z[n2] = v;
            sum2 += v;
        }
        else { z[n2] = f2{0.0f, 0.0f}; }
    }
    if (DITHER)                                                             // c/kaldi_mfcc.c:243-262: sample n is member n & 1 of Box-Muller pair n >> 1
    {
        sum2 = f2{0.0f, 0.0f};
#pragma unroll
        for (int n2 = 0; n2 < NZ; ++n2)
        {
            const unsigned i = (unsigned)(n1 + 16 * n2) >> 1;
            z[n2] += f2{dither_one(dither_bits(dkey_a, i), p.dither, n1 & 1), dither_one(dither_bits(dkey_b, i), p.dither, n1 & 1)};
            if (n2 == NZ - 1) { z[n2].x = vl ? z[n2].x : 0.0f; z[n2].y = vl ? z[n2].y : 0.0f; }
            sum2 += z[n2];
        }
    }
    f2 mean2 = {0.0f, 0.0f};
    if (p.dc0) { mean2 = f2{row16_sum(sum2.x), row16_sum(sum2.y)} * p.invL; }      // c/kaldi_mfcc.c:265-271, per frame
    const float npre = -p.preemph;
    const f2 npre2 = {npre, npre};
    const f2 *win2 = reinterpret_cast<const f2 *>(s_tab + TAB_WIN);         // (w[n], w[n]) pairs in this mode
    f2 e2 = {0.0f, 0.0f};
#pragma unroll
    for (int n2 = 0; n2 < NZ; ++n2)
    {
        z[n2] -= mean2;
        if (n2 == NZ - 1) { z[n2].x = vl ? z[n2].x : 0.0f; z[n2].y = vl ? z[n2].y : 0.0f; }
    }
    if (p.want_pre)                                                         // c/kaldi_mfcc.c:274-279
    {
#pragma unroll
        for (int n2 = 0; n2 < NZ; ++n2) { e2 = __builtin_elementwise_fma(z[n2], z[n2], e2); }
    }
    const int nprev0 = (n1 > 0) ? n1 - 1 : 0;                               // sample before n (itself for n = 0)
    if (DITHER)
    {
        // the dithered samples exist in registers only: the sample before n is the lane to the left in the same row (DPP
        // rotate within the pair's 16 lanes), for lane 0 that of lane 15 one row up
        f2 up = z[0];                                                       // n = 0: y[0] = d[0] - p d[0]
#pragma unroll
        for (int n2 = 0; n2 < NZ; ++n2)
        {
            const f2 left = {dpp_mov<0x121>(z[n2].x), dpp_mov<0x121>(z[n2].y)};     // row_ror:1
            const f2 prev = {(n1 == 0) ? up.x : left.x, (n1 == 0) ? up.y : left.y};
            up = left;
            f2 y = __builtin_elementwise_fma(npre2, prev, z[n2]);
            y *= win2[n1 + 16 * n2];
            if (n2 == NZ - 1) { y.x = vl ? y.x : 0.0f; y.y = vl ? y.y : 0.0f; }
            z[n2] = y;
        }
    }
    else
#pragma unroll
    for (int n2 = 0; n2 < NZ; ++n2)
    {
        // pre-emphasis y[n] = d[n] - p d[n-1], y[0] = d[0] - p d[0] (c/kaldi_mfcc.c:282-288), both frames at once
        const int np = (n2 == 0) ? nprev0 : n1 + 16 * n2 - 1;
        const f2 prev = f2{fa[np], fb[np]} - mean2;
        f2 y = __builtin_elementwise_fma(npre2, prev, z[n2]);
        y *= win2[n1 + 16 * n2];                                            // window: c/kaldi_mfcc.c:291 (table is zero beyond L)
        if (n2 == NZ - 1) { y.x = vl ? y.x : 0.0f; y.y = vl ? y.y : 0.0f; }
        z[n2] = y;
    }
    if (p.want_post)                                                        // c/kaldi_mfcc.c:295-301
    {
#pragma unroll
        for (int n2 = 0; n2 < NZ; ++n2) { e2 = __builtin_elementwise_fma(z[n2], z[n2], e2); }
    }
    rawe = f2{fmaxf(row16_sum(e2.x), FLT_EPSILON), fmaxf(row16_sum(e2.y), FLT_EPSILON)};

    fft256<HALF>(z, s_tab, s_tr, lane);
    const int k2 = n1;

    // partner lane (16 - k2) mod 16 holds Z[256 - k] in register 15 - k1; lane 0 is its own partner with register (16 - k1) mod 16
    const int paddr = ((lane & 48) | ((16 - k2) & 15)) << 2;
    const bool lane0 = (k2 == 0);
    {
        const f2 q8 = z[P16(8)] * z[P16(8)];                                // bin 128 (lane 0): XA = Re Z[128], XB = Im Z[128]
        pa[8] = 4.0f * q8.x;
        pb[8] = 4.0f * q8.y;
    }
#pragma unroll
    for (int k1 = 0; k1 < 8; ++k1)
    {
        const f2 zo = z[P16((15 - k1) & 15)], zs = z[P16((16 - k1) & 15)];
        f2 zp;
        zp.x = __int_as_float(__builtin_amdgcn_ds_bpermute(paddr, __float_as_int(zo.x)));
        zp.y = __int_as_float(__builtin_amdgcn_ds_bpermute(paddr, __float_as_int(zo.y)));
        zp.x = lane0 ? zs.x : zp.x;
        zp.y = lane0 ? zs.y : zp.y;
        const f2 zk = z[P16(k1)];
        const f2 e = add_conj(zk, zp);                                      // 2 XA[k]
        const f2 o = mi_sub_conj(zk, zp);                                   // 2 XB[k]
        const f2 qe = e * e, qo = o * o;
        pa[k1] = qe.x + qe.y;
        pb[k1] = qo.x + qo.y;
    }
}

// XCD-aware tile order: workgroups are dealt round-robin over the 8 XCDs, so give each XCD one contiguous run of
// tiles (neighbouring tiles share L - stp of their samples: the overlap then hits in that XCD's L2).
__device__ __forceinline__ uint32_t xcd_tile(uint32_t b, uint32_t ntiles)
{
    const uint32_t x = b & 7u, i = b >> 3, q = ntiles >> 3, r = ntiles & 7u;
    return x * q + (x < r ? x : r) + i;
}

// A zero the compiler cannot see through: added to the LDS table base inside the sub-tile loop so that the loop-invariant
// table reads (window, twiddles: 74 registers' worth) are not hoisted out of the loop at the cost of occupancy.
__device__ __forceinline__ int opaque_zero()
{
    int z = 0;
    asm volatile("" : "+s"(z));
    return z;
}

// Geometry of the quad whose first frame is tile-relative frame fq: frames, first sample, samples needed.
__device__ __forceinline__ int quad_geom(const FastParams &p, const TileDesc &tile, int fq, long long &s0, int &need, const int fw = QF)
{
    const int nfr = min((int)tile.nframes - fq, fw);
    const long long w0 = (long long)tile.frame0 + fq;
    s0 = w0 * p.stp + (p.snip_edges ? 0 : (long long)(p.stp / 2) - (long long)(p.L / 2));     // c/kaldi_mfcc.c:81-82
    need = (nfr - 1) * p.stp + p.L;
    return nfr;
}

// The next quad's audio is fetched into registers BEFORE the current quad's arithmetic and parked in the wave's LDS slice
// after it, so the HBM latency of one quad hides behind the FFT of the previous one.  Interior quads only (an edge quad
// of a snip_edges=0 utterance is loaded by load_span when its turn comes); loads past `need` return 0 (range check).
#ifndef ST_AUX
#define ST_AUX 0
#endif
template <int NP = NPRE>
struct Prefetch {
    float v[NP];                      // float samples, or the raw 16-bit PCM words (converted when parked)
    bool on;
};

// One straight run of loads whatever the case: a quad that is not prefetched gets a zero-length descriptor (its loads
// return 0 without touching memory).  Branching around the loads, or between a float and a PCM variant of them, makes the
// register allocator join the variants through copies -- and a copy of a load's destination is a wait for the load, right
// where it was issued.  Hence also PCM as a template parameter.
template <bool PCM, int NP = NPRE>
__device__ __forceinline__ void prefetch_issue(Prefetch<NP> &pf, const FastParams &p, const float *xu, const short *xu16, long long s0, int need, long long N, int nfr, int lane)
{
    pf.on = (nfr > 0) && (s0 >= 0) && (s0 + need <= N) && (need <= 64 * NP) && !(SPEECH_ABL(p) & 2);
    if (PCM)
    {
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)(xu16 + (pf.on ? s0 : 0)), 0, pf.on ? need * 2 : 0, BUF_FLAGS);
#pragma unroll
        for (int j = 0; j < NP; ++j) { pf.v[j] = __builtin_bit_cast(float, (unsigned)__builtin_amdgcn_raw_buffer_load_b16(rs, lane * 2, j * 128, 0)); }
    }
    else
    {
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)(xu + (pf.on ? s0 : 0)), 0, pf.on ? need * 4 : 0, BUF_FLAGS);
#pragma unroll
        for (int j = 0; j < NP; ++j) { pf.v[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, lane * 4, j * 256, 0)); }
    }
}

template <bool PCM, int NP = NPRE, bool PADX = false>
__device__ __forceinline__ void prefetch_park(const Prefetch<NP> &pf, float *s_aud, const float *xu, const short *xu16, long long s0, int need, long long N, int nfr, int lane, int ablate)
{
    if (pf.on)
    {
        constexpr int ROW = PADX ? 68 : 64;
        const int lofs = PADX ? lane + 2 * (lane >> 5) : lane;
#pragma unroll
        for (int j = 0; j < NP; ++j) { s_aud[lofs + ROW * j] = PCM ? (float)(short)(unsigned short)__builtin_bit_cast(unsigned, pf.v[j]) : pf.v[j]; }
    }
    else if (nfr > 0 && !(ablate & 2)) { load_span<PADX>(s_aud, xu, xu16, s0, need, N, lane); }
}

}  // namespace
}  // namespace speech
