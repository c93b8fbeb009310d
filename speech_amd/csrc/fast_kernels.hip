// fast_kernels.hip -- tuned gfx950 fp32 kernels for 512-point frames (257..512 samples per frame, even frame shift;
// dithered frames have their own instantiations) and 256-point frames (129..256 samples), two frames per complex
// transform.  Everything else runs on generic_kernels.hip.
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
#include "speech_internal.hpp"
#include <cfloat>
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <set>
#include <utility>

namespace speech {

constexpr int MAX_MTILES = 8;            // mel tiles of up to 16 columns (B <= 128), balanced over the waves; cepstral tiles of 16 (C <= 96)

struct FastGeom { bool ok, pair; int fw, nz, tile; size_t span, ws, lds; };      // launch geometry of a plan, fast_geom()

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

namespace {

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
    int do_dither;                      // c/kaldi_mfcc.c:243: dither > FLT_EPSILON
    float dither;
    unsigned long long seed;
    int ablate;                         // diagnostics only (SPEECH_ABLATE): 1 = no feature stores, 2 = no audio loads (spectrogram kernel), 4 = mel kinds stop after the FFT, 8 = stores land in a 2 MB (L2-resident) window, 16 = every utterance reads utterance 0's audio (spectrogram kernel)
    const float *tab;
    // mel kinds
    const float *melw, *mat2, *aux;
    int B, C, MP, Cp, K2, ntile_m, ntile_c;
    int amp, lg, use_energy, compress_mode;
    float compress;
    int mel_kbeg[MAX_MTILES], mel_ksteps[MAX_MTILES], mel_woff[MAX_MTILES], mel_first[MAX_MTILES], mel_cnt[MAX_MTILES];
};

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

__device__ __forceinline__ float floor_log(float v) { return pos_log(fmaxf(v, FLT_EPSILON)); }

// spectrogram: log(max(q / 4, FLT_EPSILON)) with v_log_f32 on the whole value (absolute error ~2e-6 at q ~ 1e9; the
// spectrogram tolerance is 1e-5) -- three instructions per bin
__device__ __forceinline__ float spec_log(float q)
{
    return fmaf(__log2f(fmaxf(q, 4.0f * FLT_EPSILON)), 0.69314718055994530942f, -1.38629436111989061883f);
}

// two bins at a time: the scale-and-shift is one v_pk_fma_f32 (5 instructions per pair instead of 6)
__device__ __forceinline__ f2 spec_log2(float qa, float qb)
{
    const f2 l = {__log2f(fmaxf(qa, 4.0f * FLT_EPSILON)), __log2f(fmaxf(qb, 4.0f * FLT_EPSILON))};
    return __builtin_elementwise_fma(l, f2{0.69314718055994530942f, 0.69314718055994530942f}, f2{-1.38629436111989061883f, -1.38629436111989061883f});
}

// Load the wave's audio slice: `need` samples starting at utterance-relative position s0 (may be negative or run past N
// when snip_edges is off: reflected as c/kaldi_mfcc.c:228).  All arguments but `lane` are wave-uniform.  xu16 non-null:
// the utterance is 16-bit PCM at xu16 and is converted here (exact), otherwise float at xu.
// Interior quads use raw buffer loads: one sample per lane per instruction, lane offset in one VGPR, the step in an
// SGPR, and the descriptor's range check instead of a per-lane bound test.
__device__ __forceinline__ void load_span(float *s_aud, const float *xu, const short *xu16, long long s0, int need, long long N, int lane)
{
    if (s0 >= 0 && s0 + need <= N)
    {
        const int nit = (need + 63) >> 6;
        // no per-lane bound: the slice is a whole number of 64-float rows (a guarded store here serialises the loads:
        // 0.78 vs 0.57 ms on the mfcc workload), loads past `need` return 0
        if (xu16)
        {
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)(xu16 + s0), 0, need * 2, BUF_FLAGS);
            for (int j = 0; j < nit; ++j) { s_aud[lane + 64 * j] = (float)(short)__builtin_amdgcn_raw_buffer_load_b16(rs, lane * 2, j * 128, 0); }
        }
        else
        {
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)(xu + s0), 0, need * 4, BUF_FLAGS);
            for (int j = 0; j < nit; ++j) { s_aud[lane + 64 * j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, lane * 4, j * 256, 0)); }
        }
    }
    else
    {
        for (int i = lane; i < need; i += 64)
        {
            long long n = s0 + i;
            while (n < 0 || n >= N) { n = (n < 0) ? -n - 1 : 2 * N - 1 - n; }
            s_aud[i] = xu16 ? (float)xu16[n] : xu[n];
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
template <bool HALF>
__device__ __forceinline__ void fft256(f2 (&z)[16], const float *s_tab, float *s_tr, const int lane)
{
    const int f = lane >> 4, n1 = lane & 15;
    fft16(z);
    const f2 *tw1 = reinterpret_cast<const f2 *>(s_tab + TAB_TW1);
#pragma unroll
    for (int k2 = 1; k2 < 16; ++k2) { z[P16(k2)] = cmul(z[P16(k2)], tw1[k2 * 16 + n1]); }
    wave_sync();                                                            // every lane has its samples: the slice becomes the transposition image
    const int k2 = n1;                                                      // the lane's role after the transpose
    if (HALF)
    {
        // real parts then imaginary parts (a 1 KB image per frame keeps 7 workgroups on a CU): element (k2, n) lives at
        // 16 k2 + 4 ((n>>2) ^ (k2>>2)) + (n&3), which makes the b32 writes and the b128 reads conflict-free
        float *trf = s_tr + f * 256;
        int woff[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) { woff[c] = (((n1 >> 2) ^ c) << 2) + (n1 & 3); }
        const float *rrow = trf + k2 * 16;
        const int rq = k2 >> 2;
        float4 rr[4];
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) { trf[kk * 16 + woff[kk >> 2]] = z[P16(kk)].x; }
        wave_sync();
#pragma unroll
        for (int q = 0; q < 4; ++q) { rr[q] = *reinterpret_cast<const float4 *>(rrow + ((q ^ rq) << 2)); }
        wave_sync();
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) { trf[kk * 16 + woff[kk >> 2]] = z[P16(kk)].y; }
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
    if (p.dc0) { mean = row16_sum(sum2.x + sum2.y) / (float)L; }            // c/kaldi_mfcc.c:265-271
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
        const f2 a = e + t, b = e - t;
        const f2 qa = a * a, qb = b * b;
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
    if (p.dc0) { mean2 = f2{row16_sum(sum2.x) / (float)L, row16_sum(sum2.y) / (float)L}; }      // c/kaldi_mfcc.c:265-271, per frame
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
struct Prefetch {
    float v[NPRE];                      // float samples, or the raw 16-bit PCM words (converted when parked)
    bool on;
};

// One straight run of loads whatever the case: a quad that is not prefetched gets a zero-length descriptor (its loads
// return 0 without touching memory).  Branching around the loads, or between a float and a PCM variant of them, makes the
// register allocator join the variants through copies -- and a copy of a load's destination is a wait for the load, right
// where it was issued.  Hence also PCM as a template parameter.
template <bool PCM>
__device__ __forceinline__ void prefetch_issue(Prefetch &pf, const FastParams &p, const float *xu, const short *xu16, long long s0, int need, long long N, int nfr, int lane)
{
    pf.on = (nfr > 0) && (s0 >= 0) && (s0 + need <= N) && (need <= 64 * NPRE) && !(p.ablate & 2);
    if (PCM)
    {
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)(xu16 + (pf.on ? s0 : 0)), 0, pf.on ? need * 2 : 0, BUF_FLAGS);
#pragma unroll
        for (int j = 0; j < NPRE; ++j) { pf.v[j] = __builtin_bit_cast(float, (unsigned)__builtin_amdgcn_raw_buffer_load_b16(rs, lane * 2, j * 128, 0)); }
    }
    else
    {
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)(xu + (pf.on ? s0 : 0)), 0, pf.on ? need * 4 : 0, BUF_FLAGS);
#pragma unroll
        for (int j = 0; j < NPRE; ++j) { pf.v[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, lane * 4, j * 256, 0)); }
    }
}

template <bool PCM>
__device__ __forceinline__ void prefetch_park(const Prefetch &pf, float *s_aud, const float *xu, const short *xu16, long long s0, int need, long long N, int nfr, int lane, int ablate)
{
    if (pf.on)
    {
#pragma unroll
        for (int j = 0; j < NPRE; ++j) { s_aud[lane + 64 * j] = PCM ? (float)(short)(unsigned short)__builtin_bit_cast(unsigned, pf.v[j]) : pf.v[j]; }
    }
    else if (nfr > 0 && !(ablate & 2)) { load_span(s_aud, xu, xu16, s0, need, N, lane); }
}

// PAIR (256-point plans): a wave carries 8 frames as 4 pairs (frame_power_pair), a sub-tile is 32 frames, a
// tile 128; NZ then counts rows of 16 samples.
template <int NZ, bool DITHER, bool PCM, bool PAIR>
__global__ __launch_bounds__(NT, 4) void fast_spectrogram_kernel(const FastParams p)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    float *s_tab = smem;
    float *s_aud = smem + TAB_FLOATS + wv * p.ws;                           // the wave's slice: audio, then transposition image
    float *s_tr = s_aud;
    for (int i = tid; i < TAB_FLOATS; i += NT) { s_tab[i] = p.tab[i]; }

    const TileDesc tile = p.tiles[xcd_tile(blockIdx.x, p.ntiles)];
    const UttDesc utt = p.utts[tile.utt];
    const float *xu = p.X + ((p.ablate & 16) ? 0 : utt.x_off);
    const short *xu16 = p.X16 ? p.X16 + utt.x_off : nullptr;
    const long long N = (long long)utt.n;
    constexpr int FW = PAIR ? 2 * QF : QF;                                  // frames per wave
    constexpr int SUBF = (NT / 64) * FW;                                    // frames per sub-tile
    const int nsub = ((int)tile.nframes + SUBF - 1) / SUBF;
    const unsigned long long dkey_utt = DITHER ? mix64(p.seed ^ ((unsigned long long)tile.utt << 32)) : 0ull;
    long long s0; int need;
    int nfr = quad_geom(p, tile, wv * FW, s0, need, FW);
    if (nfr > 0) { load_span(s_aud, xu, xu16, s0, need, N, lane); }
    __syncthreads();                                                        // tables (and, wave-locally, the first span) are in LDS

    const int f = lane >> 4, k2 = lane & 15;
    for (int it = 0; it < nsub; ++it)
    {
        const int fq = it * SUBF + wv * FW;                                  // this wave's quad in sub-tile `it`
        long long s0n; int needn;
        const int nfrn = quad_geom(p, tile, fq + SUBF, s0n, needn, FW);
        Prefetch pf;
        prefetch_issue<PCM>(pf, p, xu, xu16, s0n, needn, N, nfrn, lane);
        float pa[9], pb[9], rawe = 1.0f;
        f2 rawe2 = {1.0f, 1.0f};
        if (nfr > 0)
        {
            if (PAIR)
            {
                // lanes 16 f .. 16 f + 15: frames fq + 2 f (A) and fq + 2 f + 1 (B; a missing B repeats A, its outputs are dropped)
                const unsigned long long wa = (unsigned long long)((long long)tile.frame0 + fq + 2 * f);
                const unsigned long long ka = DITHER ? dither_frame_key(dkey_utt, wa) : 0ull, kb = DITHER ? dither_frame_key(dkey_utt, wa + 1) : 0ull;
                frame_power_pair<NZ, false, DITHER>(s_aud + 2 * f * p.stp, (2 * f + 1 < nfr) ? p.stp : 0, s_tab + opaque_zero(), s_tr, lane, p, ka, kb, pa, pb, rawe2);
            }
            else
            {
                const unsigned long long dkey = DITHER ? dither_frame_key(dkey_utt, (unsigned long long)((long long)tile.frame0 + fq + f)) : 0ull;
                frame_power<NZ, false, DITHER>(s_aud + f * p.stp, s_tab + opaque_zero(), s_tr, lane, p, dkey, pa, pb, rawe);
            }
        }
        // The slice is free once the transposition image has been read (inside frame_power): park the next quad now,
        // BEFORE this quad's stores are issued.  The loads were issued ahead of the whole FFT, and vmcnt retires in order,
        // so waiting for them here waits for nothing younger; the stores then drain behind the next quad's arithmetic.
        wave_sync();
        prefetch_park<PCM>(pf, s_aud, xu, xu16, s0n, needn, N, nfrn, lane, p.ablate);
        wave_sync();
        if (nfr > 0)
        {
            // c/kaldi_spectrogram.c:236-251: bin 0 := energy, bins 1..255 := power, bin 256 := bin 255, log of everything.
            // pa/pb carry 4 |X|^2: log(max(q/4, eps)) = ln2 * log2(max(q, 4 eps)) - 2 ln2.
            // The quad's 4 x 257 outputs are contiguous in Y: they are put in order in the slice behind the parked audio and
            // leave as 16-byte stores, 1 KB of whole cache lines per instruction -- the write path to HBM is bound by
            // requests in flight, and dword stores of 64-byte runs need eight times as many for the same bytes.
            const int D = p.D;
            float *st = s_aud + p.span + f * D;
            if (PAIR)
            {
                // pa[k1] / pb[k1]: bin k2 + 16 k1 of frames A / B (D = 129 columns each, B's row right behind A's)
                float *sta = s_aud + p.span + 2 * f * D, *stb = sta + D;
                {
                    const f2 l = spec_log2((k2 == 0) ? 4.0f * rawe2.x : pa[0], (k2 == 0) ? 4.0f * rawe2.y : pb[0]);
                    sta[k2] = l.x; stb[k2] = l.y;
                }
#pragma unroll
                for (int k1 = 1; k1 < 8; ++k1)
                {
                    const f2 l = spec_log2(pa[k1], pb[k1]);
                    sta[k2 + 16 * k1] = l.x; stb[k2 + 16 * k1] = l.y;
                    if (k1 == 7 && k2 == 15) { sta[128] = l.x; stb[128] = l.y; }              // last bin := the one before (c/kaldi_spectrogram.c:245-247)
                }
            }
            else
            {
            {
                const f2 l = spec_log2((k2 == 0) ? 4.0f * rawe : pa[0], pb[0]);
                st[k2] = l.x;
                if (k2 > 0) { st[NC - k2] = l.y; }
                if (k2 == 1) { st[NC] = l.y; }                                                  // last bin := the one before (c/kaldi_spectrogram.c:245-247)
            }
#pragma unroll
            for (int k1 = 1; k1 < 8; ++k1)
            {
                const f2 l = spec_log2(pa[k1], pb[k1]);
                st[k2 + 16 * k1] = l.x; st[NC - k2 - 16 * k1] = l.y;
            }
            if (k2 == 0) { st[128] = spec_log(pa[8]); }
            }
            wave_sync();
            const float *so = s_aud + p.span;
            const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(
                (void *)(p.Y + ((p.ablate & 8) ? (unsigned long long)((blockIdx.x & 31u) * TILE_SPEC + (fq & (TILE_SPEC - 1))) * D      /* diagnostics */
                                               : utt.y_off + (unsigned long long)((long long)tile.frame0 + fq) * D)), 0, (p.ablate & 1) ? 0 : nfr * D * 4, BUF_FLAGS);
            if (nfr == FW)
            {
                // FW * D floats = FW D / 4 chunks of 16 bytes (4 x 257: four full instructions and one lane; 8 x 129: 258 chunks)
                for (int c = lane; c < FW * D / 4; c += 64)
                {
                    const u32x4 v = *reinterpret_cast<const u32x4 *>(so + 4 * c);
                    __builtin_amdgcn_raw_buffer_store_b128(v, ry, c * 16, 0, 0);
                }
            }
            else                                                            // the last quad of an utterance: fewer frames
            {
                const int n = nfr * D;
                for (int i = lane; i < n; i += 64) { __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, so[i]), ry, i * 4, 0, 0); }
            }
        }
        nfr = nfrn;
    }
}

// One 16 x 16 output tile of  A[16 frames][K] . Bm[K][16]  on the matrix cores.  `arow` points at this lane's A element
// of k-step 0 (LDS, row pitch chosen so that the 64 b32 reads of a step are conflict-free), `brow` at its Bm element
// (global, L1/L2-resident; consecutive lanes read consecutive floats), `bstep` = floats between k-steps of Bm.
// v_mfma_f32_16x16x4_f32 is an exact f32 fma chain in k order; two accumulators hide its 40-cycle dependent latency.
__device__ __forceinline__ f32x4 mfma_tile(const float *arow, const float *brow, const int bstep, const int ksteps)
{
    f32x4 acc0 = {0.0f, 0.0f, 0.0f, 0.0f}, acc1 = {0.0f, 0.0f, 0.0f, 0.0f};
    int s = 0;
    for (; s + 1 < ksteps; s += 2)
    {
        const float a0 = arow[4 * s], b0 = brow[(size_t)s * bstep];
        const float a1 = arow[4 * s + 4], b1 = brow[(size_t)(s + 1) * bstep];
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b0, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b1, acc1, 0, 0, 0);
    }
    if (s < ksteps) { acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(arow[4 * s], brow[(size_t)s * bstep], acc0, 0, 0, 0); }
    return acc0 + acc1;
}

// The mel contraction: `ksteps` is a multiple of 4 (host pads with zero weights).  Operands of the next four k-steps are
// loaded while the current four MFMAs run, so the L1/L2 latency of the weights is paid once per tile instead of per step
// (8 more VGPRs; fetching deeper costs occupancy and measured slower).
__device__ __forceinline__ f32x4 mfma_tile4(const float *arow, const __amdgpu_buffer_rsrc_t rb, const int bvoff, const int ksteps)
{
    // the weights come through a raw buffer descriptor: the lane's byte offset stays in one VGPR and the k-step advances
    // in an SGPR, so the loop spends no vector instructions on 64-bit addresses
    f32x4 acc0 = {0.0f, 0.0f, 0.0f, 0.0f}, acc1 = {0.0f, 0.0f, 0.0f, 0.0f};
    if (ksteps <= 0) { return acc0; }
    float a[4], b[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { b[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rb, bvoff, 256 * j, 0)); a[j] = arow[4 * j]; }
    for (int s = 0; s < ksteps; s += 4)
    {
        const int sn = min(s + 4, ksteps - 4);                              // the last trip reloads its own block (harmless)
        const float *an_p = arow + 4 * sn;
        float an[4], bn[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) { bn[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rb, bvoff, 256 * (sn + j), 0)); an[j] = an_p[4 * j]; }
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], b[0], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[1], b[1], acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2], b[2], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[3], b[3], acc1, 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 4; ++j) { a[j] = an[j]; b[j] = bn[j]; }
    }
    return acc0 + acc1;
}

// fbank / mfcc / plp.  KIND = SPEECH_FBANK | SPEECH_MFCC | SPEECH_PLP;  P = LPC order (plp only, compile-time so that
// Levinson-Durbin and the cepstral recursion stay in registers).
// PAIR (256-point plans): 8 frames per wave as 4 pairs (frame_power_pair), 64 frames per tile, a 64 x 130
// power tile; NZ then counts rows of 16 samples.
template <int NZ, int KIND, int P, bool DITHER, bool PAIR>
__global__ __launch_bounds__(NTM, 6) void fast_mel_kernel(const FastParams p)      // 6 waves per SIMD = 3 workgroups per CU: at most 80 VGPRs
{
    constexpr int FWK = PAIR ? 2 * QF : QF;                                 // frames per wave
    constexpr int TILEK = MW * FWK;                                         // frames per workgroup
    constexpr int RTK = TILEK / 16;                                         // 16-frame row tiles
    constexpr int PPK = PAIR ? PP2 : PP;                                     // pitch of the power tile
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    float *s_tab = smem;
    float *s_aud = smem + TAB_FLOATS + wv * p.ws;                           // the wave's slice: audio, then transposition image
    float *s_tr = s_aud;
    // after the FFTs the slices are dead and hold the TILEK x PPK power tile
    float *s_pow = smem + TAB_FLOATS;
    // private region behind the slices: frame energies | fbank: output tile | mfcc, plp: mel tile.  The mfcc / plp output
    // tile (and plp's autocorrelations) are written after the power tile is dead and reuse its place.
    float *s_e = smem + TAB_FLOATS + MW * p.ws;
    float *s_out = (KIND == SPEECH_FBANK) ? s_e + TILEK : s_pow;
    float *s_mel = s_e + TILEK;
    float *s_ac = s_out + ((TILEK * p.D + 3) & ~3);
    for (int i = tid; i < TAB_FLOATS; i += NTM) { s_tab[i] = p.tab[i]; }

    const TileDesc tile = p.tiles[xcd_tile(blockIdx.x, p.ntiles)];
    const UttDesc utt = p.utts[tile.utt];
    const int fq = wv * FWK;
    const int nfr = min((int)tile.nframes - fq, FWK);
    if (nfr > 0)
    {
        const long long w0 = (long long)tile.frame0 + fq;
        const long long s0 = w0 * p.stp + (p.snip_edges ? 0 : (long long)(p.stp / 2) - (long long)(p.L / 2));   // c/kaldi_mfcc.c:81-82
        load_span(s_aud, p.X + utt.x_off, p.X16 ? p.X16 + utt.x_off : nullptr, s0, (nfr - 1) * p.stp + p.L, (long long)utt.n, lane);
    }
    __syncthreads();

    const int f = lane >> 4, k2 = lane & 15;
    float pa[9], pb[9], rawe = 1.0f;
    f2 rawe2 = {1.0f, 1.0f};
    if (nfr > 0 && PAIR)
    {
        // lanes 16 f .. 16 f + 15: frames fq + 2 f (A) and fq + 2 f + 1 (B; a missing B repeats A, its row is never stored)
        const unsigned long long ku = DITHER ? mix64(p.seed ^ ((unsigned long long)tile.utt << 32)) : 0ull;
        const unsigned long long wa = (unsigned long long)((long long)tile.frame0 + fq + 2 * f);
        const unsigned long long ka = DITHER ? dither_frame_key(ku, wa) : 0ull, kb = DITHER ? dither_frame_key(ku, wa + 1) : 0ull;
        frame_power_pair<NZ, true, DITHER>(s_aud + 2 * f * p.stp, (2 * f + 1 < nfr) ? p.stp : 0, s_tab, s_tr, lane, p, ka, kb, pa, pb, rawe2);
    }
    else if (nfr > 0)
    {
        const unsigned long long dkey = DITHER ? dither_frame_key(mix64(p.seed ^ ((unsigned long long)tile.utt << 32)),
                                                                  (unsigned long long)((long long)tile.frame0 + fq + f)) : 0ull;
        frame_power<NZ, true, DITHER>(s_aud + f * p.stp, s_tab, s_tr, lane, p, dkey, pa, pb, rawe);
    }
    else
    {
#pragma unroll
        for (int i = 0; i < 9; ++i) { pa[i] = 0.0f; pb[i] = 0.0f; }
    }
    // The power tile takes the slices' place.  When a slice is exactly as long as the wave's rows of the tile (the usual
    // case: ws = FWK * PPK), a wave's rows lie in its own slice and it only has to be done with its own transposition image.
    if (p.ws != FWK * PPK) { __syncthreads(); } else { wave_sync(); }
    if (p.ablate & 4) { if (pa[0] + pb[3] + rawe + rawe2.x == 123.456f) { p.Y[tid] = pa[1]; } return; }      // diagnostics: front end + FFT only

    // power (c/kaldi_mfcc.c:307-310; amplitude if amp, c/kaldi_fbank.c:312-316) -> TILEK x 257 tile, column 257 zero
    if (PAIR)
    {
        // pa[k1] / pb[k1]: bin k2 + 16 k1 of frames A / B, 129 bins each; column 129 is the zero a padded k-loop may touch
        float *pra = s_pow + (fq + 2 * f) * PPK, *prb = pra + PPK;
        if (p.amp)
        {
#pragma unroll
            for (int k1 = 0; k1 < 9; ++k1) { pa[k1] = __fsqrt_rn(pa[k1]); pb[k1] = __fsqrt_rn(pb[k1]); }
        }
#pragma unroll
        for (int k1 = 0; k1 < 8; ++k1)
        {
            pra[k2 + 16 * k1] = pa[k1];
            prb[k2 + 16 * k1] = pb[k1];
        }
        if (k2 == 0)
        {
            pra[128] = pa[8]; prb[128] = pb[8]; pra[129] = 0.0f; prb[129] = 0.0f;
            s_e[fq + 2 * f] = rawe2.x; s_e[fq + 2 * f + 1] = rawe2.y;
        }
    }
    else
    {
        float *pr = s_pow + (fq + f) * PPK;
        if (p.amp)
        {
#pragma unroll
            for (int k1 = 0; k1 < 9; ++k1) { pa[k1] = __fsqrt_rn(pa[k1]); pb[k1] = __fsqrt_rn(pb[k1]); }
        }
#pragma unroll
        for (int k1 = 0; k1 < 8; ++k1)
        {
            pr[k2 + 16 * k1] = pa[k1];
            pr[NC - k2 - 16 * k1] = pb[k1];
        }
        if (k2 == 0) { pr[128] = pa[8]; pr[NC + 1] = 0.0f; s_e[fq + f] = rawe; }       // column 257: the zero a padded k-loop may touch
    }
    __syncthreads();

    // mel filterbank: frames x bins . bins x mels, banded (c/kaldi_mfcc.c:313-320, c/kaldi_fbank.c:337-345)
    const int eo = (KIND == SPEECH_FBANK && p.use_energy) ? 1 : 0;
    const int fr0 = (lane >> 4) * 4, col = lane & 15;
    for (int job = wv; job < RTK * p.ntile_m; job += MW)
    {
        const int t = job / RTK, rt = job - t * RTK;                          // mel tile, 16-frame row tile
        const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void *)(p.melw + p.mel_woff[t]), 0, 256 * p.mel_ksteps[t], BUF_FLAGS);
        const f32x4 acc = mfma_tile4(s_pow + (16 * rt + col) * PPK + p.mel_kbeg[t] + (lane >> 4), rb, lane * 4, p.mel_ksteps[t]);
        const int m = p.mel_first[t] + col;                                 // this lane's mel bin; the tile holds mel_cnt[t] <= 16 of them
        const bool live = col < p.mel_cnt[t];
#pragma unroll
        for (int r = 0; r < 4; ++r)
        {
            const int fr = 16 * rt + fr0 + r;
            const float v = acc[r];
            if (KIND == SPEECH_FBANK)
            {
                if (live) { s_out[fr * p.D + eo + m] = p.lg ? floor_log(v) : v; }           // c/kaldi_fbank.c:345 (last-bin quirk: duplicated matrix column)
            }
            else if (KIND == SPEECH_MFCC)
            {
                if (live) { s_mel[fr * p.MP + m] = floor_log(v); }                          // c/kaldi_mfcc.c:318
            }
            else
            {
                // equal loudness, compression, end duplication (c/kaldi_plp.c:339-358)
                if (live)
                {
                    const float x = v * p.aux[m];
                    const float c = (p.compress_mode == 1) ? x : ((p.compress_mode == 2) ? __fsqrt_rn(x) : __builtin_amdgcn_exp2f(p.compress * __log2f(x)));
                    s_mel[fr * p.MP + 1 + m] = c;
                    if (m == 0) { s_mel[fr * p.MP] = c; }
                    if (m == p.B - 1) { for (int j = p.B + 1; j < p.K2; ++j) { s_mel[fr * p.MP + j] = (j == p.B + 1) ? c : 0.0f; } }
                }
            }
        }
    }
    if (KIND == SPEECH_FBANK && eo && tid < TILEK) { s_out[tid * p.D] = pos_log(s_e[tid]); }     // c/kaldi_fbank.c:299
    if (KIND == SPEECH_MFCC && tid < 4 * TILEK) { const int j = p.B + (tid & 3); if (j < p.K2) { s_mel[(tid >> 2) * p.MP + j] = 0.0f; } }      // k-padding of the DCT operand
    __syncthreads();

    if (KIND == SPEECH_MFCC)
    {
        // DCT-II x lifter as a second contraction (c/kaldi_mfcc.c:323-333); the last-cepstrum quirk is a duplicated column
        for (int job = wv; job < RTK * p.ntile_c; job += MW)
        {
            const int t = job / RTK, rt = job - t * RTK;
            const f32x4 acc = mfma_tile(s_mel + (16 * rt + col) * p.MP + (lane >> 4), p.mat2 + (lane >> 4) * p.Cp + 16 * t + col, 4 * p.Cp, p.K2 >> 2);
            const int c = 16 * t + col;
#pragma unroll
            for (int r = 0; r < 4; ++r)
            {
                const int fr = 16 * rt + fr0 + r;
                if (c < p.C) { s_out[fr * p.C + c] = (p.use_energy && c == 0) ? pos_log(s_e[fr]) : acc[r]; }
            }
        }
        __syncthreads();
    }
    if (KIND == SPEECH_PLP)
    {
        // autocorrelation = IDFT bases x compressed spectrum, one MFMA tile per 16 frames
        if (wv < RTK)
        {
            const f32x4 acc = mfma_tile(s_mel + (16 * wv + col) * p.MP + (lane >> 4), p.mat2 + lane, 64, p.K2 >> 2);
#pragma unroll
            for (int r = 0; r < 4; ++r) { s_ac[(16 * wv + fr0 + r) * 17 + col] = acc[r]; }
        }
        __syncthreads();
        if (tid < TILEK)
        {
            // one lane per frame: Levinson-Durbin with Kaldi's 1 - k^2 >= 1e-5 guard, LPC -> cepstrum, lifter
            // (reference intent c/kaldi_plp.c:206,380-403)
            const float *r = s_ac + tid * 17;
            float a[P > 0 ? P : 1], tmp[P > 0 ? P : 1], cep[P > 0 ? P : 1];
            float E = r[0];
#pragma unroll
            for (int i = 0; i < P; ++i)
            {
                float ki = r[i + 1];
#pragma unroll
                for (int j = 0; j < i; ++j) { ki += a[j] * r[i - j]; }
                ki = ki / E;
                float c = 1.0f - ki * ki;
                c = (c < 1.0e-5f) ? 1.0e-5f : c;
                E *= c;
                tmp[i] = -ki;
#pragma unroll
                for (int j = 0; j < i; ++j) { tmp[j] = a[j] - ki * a[i - j - 1]; }
#pragma unroll
                for (int j = 0; j <= i; ++j) { a[j] = tmp[j]; }
            }
#pragma unroll
            for (int i = 0; i < P; ++i)
            {
                float sum = 0.0f;
#pragma unroll
                for (int j = 0; j < i; ++j) { sum += (float)(i - j) * a[j] * cep[i - j - 1]; }
                cep[i] = -a[i] - sum / (float)(i + 1);
            }
            const float *lift = p.aux + p.B;
            float *y = s_out + tid * p.C;
            const int C = p.C;
            float prev = 0.0f;
#pragma unroll
            for (int k = 0; k <= P; ++k)
            {
                if (k < C)
                {
                    float v = (k == 0) ? floor_log(E) : cep[k > 0 ? k - 1 : 0];
                    v *= lift[k];
                    if (k == 0 && p.use_energy) { v = pos_log(s_e[tid]); }     // c/kaldi_plp.c:397
                    if (k == C - 1 && C >= 2) { v = prev; }                                               // c/kaldi_plp.c:403
                    y[k] = v; prev = v;
                }
            }
            for (int k = P + 1; k < C; ++k) { const float v = (k == C - 1) ? prev : 0.0f; y[k] = v; prev = v; }
        }
        __syncthreads();
    }

    // the tile's nframes x D outputs are one contiguous run of Y
    float *y = p.Y + utt.y_off + (unsigned long long)tile.frame0 * (unsigned long long)p.D;
    const int nout = (int)tile.nframes * p.D;
    for (int i = tid; i < nout; i += NTM) { y[i] = s_out[i]; }
}

typedef void (*kernel_fn)(const FastParams);

template <int NZ, bool DITHER, bool PAIR>
kernel_fn kernel_for_kind(int kind, bool pcm)
{
    switch (kind)
    {
        case SPEECH_SPECTROGRAM: return pcm ? fast_spectrogram_kernel<NZ, DITHER, true, PAIR> : fast_spectrogram_kernel<NZ, DITHER, false, PAIR>;
        case SPEECH_FBANK: return fast_mel_kernel<NZ, SPEECH_FBANK, 0, DITHER, PAIR>;
        case SPEECH_MFCC: return fast_mel_kernel<NZ, SPEECH_MFCC, 0, DITHER, PAIR>;
        default: return fast_mel_kernel<NZ, SPEECH_PLP, 12, DITHER, PAIR>;
    }
}

// dithered frames have their own instantiations: the generator's temporaries would otherwise cost the plain kernels
// 13-24 VGPRs and a resident workgroup per CU
template <int NZ>
kernel_fn kernel_for_nz(int kind, bool dither, bool pcm) { return dither ? kernel_for_kind<NZ, true, false>(kind, pcm) : kernel_for_kind<NZ, false, false>(kind, pcm); }

// two frames per transform (256-point plans): NZ = ceil(L / 16) = 9 .. 16
template <int NZ>
kernel_fn kernel_for_pair_nz(int kind, bool dither, bool pcm) { return dither ? kernel_for_kind<NZ, true, true>(kind, pcm) : kernel_for_kind<NZ, false, true>(kind, pcm); }

kernel_fn kernel_for_pair(int nz, int kind, bool dither, bool pcm)
{
    switch (nz)
    {
        case 9: return kernel_for_pair_nz<9>(kind, dither, pcm);
        case 10: return kernel_for_pair_nz<10>(kind, dither, pcm);
        case 11: return kernel_for_pair_nz<11>(kind, dither, pcm);
        case 12: return kernel_for_pair_nz<12>(kind, dither, pcm);
        case 13: return kernel_for_pair_nz<13>(kind, dither, pcm);
        case 14: return kernel_for_pair_nz<14>(kind, dither, pcm);
        case 15: return kernel_for_pair_nz<15>(kind, dither, pcm);
        default: return kernel_for_pair_nz<16>(kind, dither, pcm);
    }
}

kernel_fn kernel_for(int nz, int kind, bool dither, bool pcm)
{
    switch (nz)
    {
        case 9: return kernel_for_nz<9>(kind, dither, pcm);
        case 10: return kernel_for_nz<10>(kind, dither, pcm);
        case 11: return kernel_for_nz<11>(kind, dither, pcm);
        case 12: return kernel_for_nz<12>(kind, dither, pcm);
        case 13: return kernel_for_nz<13>(kind, dither, pcm);
        case 14: return kernel_for_nz<14>(kind, dither, pcm);
        case 15: return kernel_for_nz<15>(kind, dither, pcm);
        default: return kernel_for_nz<16>(kind, dither, pcm);
    }
}

int mel_pitch(int cols) { return 32 * ((cols + 31) / 32) + 2; }              // = 2 (mod 32): the MFMA A-operand reads are conflict-free

}  // namespace

// Launch geometry of a plan on the tuned kernels: 512-point frames (257..512 samples), or 256-point frames (129..256) two
// per transform (`pair`: 8 frames per wave).
static FastGeom fast_geom(const speech_opts &o, size_t L, size_t nfft)
{
    FastGeom g; memset(&g, 0, sizeof(g));
    if (o.dtype != SPEECH_F32 || (nfft != (size_t)NFFT && nfft != (size_t)NFFT / 2u)) { return g; }
    const size_t stp = (size_t)((float)o.sr * (float)o.frame_shift / 1000.0f);
    if (stp < 2u || stp > 1024u || (stp & 1u) || L > nfft || L <= nfft / 2u) { return g; }
    g.pair = (nfft == (size_t)NFFT / 2u);
    g.fw = g.pair ? 2 * QF : QF;
    g.nz = g.pair ? (int)((L + 15u) / 16u) : (int)((L + 31u) / 32u);
    const size_t F = nfft / 2u + 1u, rows = g.pair ? 16u * (size_t)g.nz : 32u * (size_t)g.nz;
    g.span = (((size_t)g.fw - 1u) * stp + rows + 63u) & ~(size_t)63;        // whole 64-lane load instructions fit the slice
    if (o.kind == SPEECH_SPECTROGRAM)
    {
        g.ws = std::max(g.span + (size_t)g.fw * F, (size_t)TR_WAVE_C);      // + the wave's output, staged behind the parked audio
        g.tile = (NT / 64) * g.fw * 4;                                      // four sub-tiles
        g.lds = (TAB_FLOATS + (NT / 64) * g.ws) * sizeof(float);
        g.ok = g.lds <= 64u * 1024u;
        return g;
    }
    if (o.B < 2u || o.B > 16u * MAX_MTILES) { return g; }                   // 128 mel bins
    size_t D = o.B + (o.use_energy ? 1u : 0u), cols = 0;
    if (o.kind == SPEECH_MFCC) { if (o.C > 96u) { return g; } D = o.C; cols = (o.B + 3u) & ~(size_t)3; }
    if (o.kind == SPEECH_PLP)
    {
        const size_t P = (o.lpc_order > 0u) ? o.lpc_order : o.C - 1u;
        if (P != 12u) { return g; }                                         // the in-register Levinson-Durbin is built for order 12
        D = o.C; cols = (o.B + 2u + 3u) & ~(size_t)3;
    }
    g.tile = MW * g.fw;
    const size_t T = (size_t)g.tile;
    g.ws = std::max(g.span, g.pair ? (T * PP2 + MW - 1) / MW : (size_t)TR_WAVE);     // the slices together hold the power tile
    // behind the slices: frame energies | fbank: output tile | mfcc, plp: mel tile; the mfcc / plp output and
    // autocorrelation tiles reuse the power tile's place
    const size_t extra = T + (o.kind == SPEECH_FBANK ? ((T * D + 3u) & ~(size_t)3) : T * (size_t)mel_pitch((int)cols));
    if (((T * D + 3u) & ~(size_t)3) + T * 17u > (size_t)MW * g.ws) { return g; }
    g.lds = (TAB_FLOATS + MW * g.ws + extra) * sizeof(float);
    g.ok = g.lds <= 64u * 1024u;
    return g;
}

bool fast_supported(const speech_opts &o, size_t L, size_t nfft) { return fast_geom(o, L, nfft).ok; }

int fast_tile_frames(const speech_opts &o, size_t L, size_t nfft) { return fast_geom(o, L, nfft).tile; }

// Opt a kernel in to 64 KB of dynamic LDS (fast_geom admits nothing larger), once per device.  Done when the plan is
// created, so that speech_batch_run itself only launches (and can be captured in a HIP graph).
static hipError_t opt_in_lds(kernel_fn k)
{
    static std::mutex mu;
    static std::set<std::pair<int, const void *>> opted;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) { return e; }
    std::lock_guard<std::mutex> lock(mu);
    const auto key = std::make_pair(dev, reinterpret_cast<const void *>(k));
    if (opted.insert(key).second)
    {
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
        if (e != hipSuccess) { opted.erase(key); }
    }
    return e;
}

static kernel_fn plan_kernel(const FastTables *t, bool dither, bool pcm)
{
    return t->geom.pair ? kernel_for_pair(t->nz, t->kind, dither, pcm) : kernel_for(t->nz, t->kind, dither, pcm);
}

static hipError_t upload_floats(const std::vector<float> &h, float **d)
{
    *d = nullptr;
    if (h.empty()) { return hipSuccess; }
    hipError_t e = hipMalloc((void **)d, h.size() * sizeof(float));
    if (e == hipSuccess) { e = hipMemcpy(*d, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice); }
    return e;
}

// Returns nullptr with *err empty when the option set turns out not to fit the tuned kernels (the plan then uses the
// generic kernels), nullptr with a message on a real failure.
FastTables *fast_tables_create(const speech_opts &o, size_t L, size_t, size_t nfft, size_t B, size_t C, size_t P, const std::vector<float> &win,
                               const std::vector<float> &f2b, const std::vector<int> &beg, const std::vector<int> &num, const std::vector<float> &dct,
                               const std::vector<float> &lift, const std::vector<float> &eql, const std::vector<float> &idft, std::string *err)
{
    if (err) { err->clear(); }
    std::vector<float> tab(TAB_FLOATS, 0.0f);
    const FastGeom g = fast_geom(o, L, nfft);
    if (!g.ok) { return nullptr; }
    if (g.pair) { for (size_t l = 0; l < L; ++l) { tab[TAB_WIN + 2 * l] = win[l]; tab[TAB_WIN + 2 * l + 1] = win[l]; } }      // (w, w): one factor for frame A and B
    else { for (size_t l = 0; l < L && l < (size_t)NFFT; ++l) { tab[TAB_WIN + l] = win[l]; } }
    const long double pi = 3.141592653589793238462643383279502884L;
    for (int k2 = 0; k2 < 16; ++k2) { for (int n1 = 0; n1 < 16; ++n1) {
        const long double a = -2.0L * pi * (long double)(n1 * k2) / (long double)NC;
        tab[TAB_TW1 + (k2 * 16 + n1) * 2] = (float)cosl(a); tab[TAB_TW1 + (k2 * 16 + n1) * 2 + 1] = (float)sinl(a); } }
    for (int k1 = 0; k1 < 8; ++k1) { for (int k2 = 0; k2 < 16; ++k2) {
        const long double a = -pi * (long double)(k2 + 16 * k1) / (long double)NC;
        tab[TAB_TW2 + (k1 * 16 + k2) * 2] = (float)cosl(a); tab[TAB_TW2 + (k1 * 16 + k2) * 2 + 1] = (float)sinl(a); } }
    FastTables *t = new (std::nothrow) FastTables();
    if (!t) { if (err) { *err = "out of host memory"; } return nullptr; }
    t->d_tab = t->d_melw = t->d_mat2 = t->d_aux = nullptr;
    t->nz = g.nz; t->kind = o.kind; t->geom = g;
    t->B = t->C = t->P = t->MP = t->Cp = t->K2 = t->ntile_m = t->ntile_c = 0;
    t->F = (int)(nfft / 2u + 1u);
    t->D = t->F;
    const int klim = g.pair ? PP2 : PP;                                      // power-tile columns a k-loop may touch
    std::vector<float> melw, mat2, aux;
    if (o.kind != SPEECH_SPECTROGRAM)
    {
        const int Bi = (int)B, Ci = (int)C;
        t->B = Bi;
        // the kernels hand over 4 |X|^2 (2 |X| with amp): the exact power-of-two factor is folded into the weights
        const float wscale = (o.kind == SPEECH_FBANK && o.amp) ? 0.5f : 0.25f;
        std::vector<int> off(B, 0);
        for (size_t b = 1; b < B; ++b) { off[b] = off[b - 1] + num[b - 1]; }
        // effective source bin of each column: fbank's last column is a copy of the one before (c/kaldi_fbank.c:350)
        auto src_of = [&](int m) { return (o.kind == SPEECH_FBANK && m == Bi - 1 && Bi >= 2) ? Bi - 2 : m; };
        // k-range of columns [i, j): k-steps in whole groups of four (mfma_tile4); the extra bins carry zero weights
        auto k_range = [&](int i, int j, int &kb, int &ks) {
            int kmin = 1 << 30, kmax = -1;
            for (int m = i; m < j; ++m) { const int b = src_of(m); if (num[b] > 0) { kmin = std::min(kmin, beg[b]); kmax = std::max(kmax, beg[b] + num[b]); } }
            ks = 0; kb = 0;
            if (kmax > kmin) { ks = (((kmax - kmin + 3) / 4) + 3) & ~3; kb = kmin; if (kb + 4 * ks > klim) { kb = klim - 4 * ks; } }
        };
        // Columns are dealt to tiles of at most 16 so that the LONGEST k-loop is as short as possible: with equal column
        // counts the high tiles (wide triangles) take 2-3x the k-steps of the low ones and their waves finish last.
        // 4 tiles (one per wave) up to 64 columns, else 8.
        const int T = (Bi <= 64) ? std::min(4, Bi) : 8;
        std::vector<std::vector<int>> best(T + 1, std::vector<int>(Bi + 1, 1 << 30)), from(T + 1, std::vector<int>(Bi + 1, -1));
        best[0][0] = 0;
        for (int tl = 1; tl <= T; ++tl) { for (int j = tl; j <= Bi; ++j) { for (int i = std::max(tl - 1, j - 16); i < j; ++i) {
            if (best[tl - 1][i] >= (1 << 30)) { continue; }
            int kb, ks; k_range(i, j, kb, ks);
            const int c = std::max(best[tl - 1][i], ks);
            if (c < best[tl][j]) { best[tl][j] = c; from[tl][j] = i; } } } }
        if (best[T][Bi] >= (1 << 30)) { delete t; return nullptr; }
        t->ntile_m = T;
        { int j = Bi; for (int tl = T; tl >= 1; --tl) { const int i = from[tl][j]; t->mel_first[tl - 1] = i; t->mel_cnt[tl - 1] = j - i; j = i; } }
        for (int tl = 0; tl < t->ntile_m; ++tl)
        {
            int kb, ks; k_range(t->mel_first[tl], t->mel_first[tl] + t->mel_cnt[tl], kb, ks);
            if (kb < 0) { delete t; return nullptr; }                       // a tile spanning more than 256 bins: generic kernels
            t->mel_kbeg[tl] = kb; t->mel_ksteps[tl] = ks; t->mel_woff[tl] = (int)melw.size();
            melw.resize(melw.size() + (size_t)64 * ks, 0.0f);
            float *w = melw.data() + t->mel_woff[tl];
            for (int c = 0; c < t->mel_cnt[tl]; ++c)
            {
                const int b = src_of(t->mel_first[tl] + c);
                for (int j = 0; j < num[b]; ++j) { w[(beg[b] + j - kb) * 16 + c] = wscale * f2b[off[b] + j]; }
            }
        }
        if (melw.empty()) { melw.push_back(0.0f); }
        t->D = Bi + (o.use_energy ? 1 : 0);
        if (o.kind == SPEECH_MFCC)
        {
            // M[b][c] = lift[c] * 2 cos(pi (b + 1/2) c / B); column C-1 := column C-2 (c/kaldi_mfcc.c:333)
            t->C = Ci; t->D = Ci; t->ntile_c = (Ci + 15) / 16; t->Cp = 16 * t->ntile_c; t->K2 = (Bi + 3) & ~3;
            mat2.assign((size_t)t->K2 * t->Cp, 0.0f);
            for (int b = 0; b < Bi; ++b) { for (int c = 0; c < Ci; ++c) {
                const int ce = (c == Ci - 1 && Ci >= 2) ? Ci - 2 : c;
                mat2[(size_t)b * t->Cp + c] = lift[ce] * dct[(size_t)ce * B + b]; } }
        }
        if (o.kind == SPEECH_PLP)
        {
            t->C = Ci; t->D = Ci; t->P = (int)P; t->K2 = (Bi + 2 + 3) & ~3;
            mat2.assign((size_t)t->K2 * 16, 0.0f);
            for (int i = 0; i <= (int)P && i < 16; ++i) { for (int j = 0; j < Bi + 2; ++j) { mat2[(size_t)j * 16 + i] = idft[(size_t)i * (B + 2) + j]; } }
            aux.assign(eql.begin(), eql.end());
            aux.insert(aux.end(), lift.begin(), lift.end());
        }
        t->MP = mel_pitch(t->K2);
    }
    hipError_t e = upload_floats(tab, &t->d_tab);
    if (e == hipSuccess) { e = upload_floats(melw, &t->d_melw); }
    if (e == hipSuccess) { e = upload_floats(mat2, &t->d_mat2); }
    if (e == hipSuccess) { e = upload_floats(aux, &t->d_aux); }
    if (e == hipSuccess && g.lds > 48u * 1024u)
    {
        const bool dither = (float)o.dither > FLT_EPSILON;
        e = opt_in_lds(plan_kernel(t, dither, false));
        if (e == hipSuccess) { e = opt_in_lds(plan_kernel(t, dither, true)); }
    }
    if (e != hipSuccess) { if (err) { *err = hipGetErrorString(e); } fast_tables_destroy(t); return nullptr; }
    return t;
}

void fast_tables_destroy(FastTables *t)
{
    if (!t) { return; }
    if (t->d_tab) { (void)hipFree(t->d_tab); }
    if (t->d_melw) { (void)hipFree(t->d_melw); }
    if (t->d_mat2) { (void)hipFree(t->d_mat2); }
    if (t->d_aux) { (void)hipFree(t->d_aux); }
    delete t;
}

size_t fast_scratch_elems(const FastTables *, size_t) { return 0; }

hipError_t launch_fast(const FastTables *t, const FrameParams<float> &fp, float *, hipStream_t s, hipEvent_t ev0, hipEvent_t ev1)
{
    if (fp.ntiles == 0) { return hipSuccess; }
    FastParams p;
    memset(&p, 0, sizeof(p));
    p.X = fp.X; p.X16 = fp.X16; p.Y = fp.Y; p.utts = fp.utts; p.tiles = fp.tiles; p.ntiles = fp.ntiles;
    p.L = fp.L; p.stp = fp.stp; p.D = fp.D;
    const FastGeom &g = t->geom;
    p.span = (int)g.span; p.ws = (int)g.ws;
    p.snip_edges = fp.snip_edges; p.dc0 = fp.dc0; p.want_pre = fp.want_pre; p.want_post = fp.want_post;
    p.preemph = fp.do_preemph ? fp.preemph : 0.0f;
    p.do_dither = fp.do_dither; p.dither = fp.dither; p.seed = fp.seed;
    { const char *ab = getenv("SPEECH_ABLATE"); p.ablate = ab ? atoi(ab) : 0; }     // timing diagnostics, DESIGN.md 4.3
    p.tab = t->d_tab;
    p.melw = t->d_melw; p.mat2 = t->d_mat2; p.aux = t->d_aux;
    p.B = t->B; p.C = t->C; p.MP = t->MP; p.Cp = t->Cp; p.K2 = t->K2; p.ntile_m = t->ntile_m; p.ntile_c = t->ntile_c;
    p.amp = fp.amp; p.lg = fp.lg; p.use_energy = fp.use_energy; p.compress_mode = fp.compress_mode; p.compress = fp.compress;
    for (int i = 0; i < MAX_MTILES; ++i) { p.mel_kbeg[i] = t->mel_kbeg[i]; p.mel_ksteps[i] = t->mel_ksteps[i]; p.mel_woff[i] = t->mel_woff[i]; p.mel_first[i] = t->mel_first[i]; p.mel_cnt[i] = t->mel_cnt[i]; }
    const bool mel = t->kind != SPEECH_SPECTROGRAM;
    const size_t lds = g.lds;
    kernel_fn k = plan_kernel(t, p.do_dither != 0, p.X16 != nullptr);
    hipError_t e = hipSuccess;
    if (ev0) { e = hipEventRecord(ev0, s); if (e != hipSuccess) { return e; } }
    hipLaunchKernelGGL(k, dim3(fp.ntiles), dim3(mel ? NTM : NT), lds, s, p);
    e = hipGetLastError();
    if (e != hipSuccess) { return e; }
    if (ev1) { e = hipEventRecord(ev1, s); }
    return e;
}

}  // namespace speech
