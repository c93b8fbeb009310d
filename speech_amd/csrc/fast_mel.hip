// fast_mel.hip -- the tuned fbank / mfcc / plp kernel: front end + FFT (fast_common.hpp), then the mel filterbank, the
// DCT or the IDFT as banded f32 MFMA contractions over the tile's power block, and the epilogues.
#include "fast_common.hpp"

namespace speech {
namespace {

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

// The same contractions with the B operand (weights) fetched AHEAD: all of a job's first KP k-steps are requested in one
// run of buffer loads, issued before the workgroup barrier that precedes the contraction, so the L2 latency of the weights
// is spent waiting at that barrier instead of once per four k-steps inside the loop.  Loads beyond the tile's k-steps fall
// outside the descriptor's range and return 0 without touching memory.  During these phases the FFT's 32 registers of
// samples are dead, so the KP registers cost no occupancy.
constexpr int KPRE = 24;

template <int KP>
__device__ __forceinline__ void weights_issue(float (&b)[KP], const __amdgpu_buffer_rsrc_t rb, const int voff, const int stepbytes)
{
#pragma unroll
    for (int j = 0; j < KP; ++j) { b[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rb, voff, stepbytes * j, 0)); }
}

// k-steps 0 .. min(ksteps, KP) - 1 from the prefetched weights, in groups of G (ksteps a multiple of G); the A operands of the
// next group are read from LDS while the current group's MFMAs run
template <int KP, int G>
__device__ __forceinline__ f32x4 mfma_tile_pre(const float *arow, const float (&b)[KP], const int ksteps)
{
    f32x4 acc0 = {0.0f, 0.0f, 0.0f, 0.0f}, acc1 = {0.0f, 0.0f, 0.0f, 0.0f};
    if (ksteps <= 0) { return acc0; }
    float a[G];
#pragma unroll
    for (int g = 0; g < G; ++g) { a[g] = arow[4 * g]; }
#pragma unroll
    for (int s = 0; s < KP; s += G)
    {
        if (s < ksteps)
        {
            const int sn = min(s + G, ksteps - G);                          // the last group reloads itself (harmless)
            float an[G];
#pragma unroll
            for (int g = 0; g < G; ++g) { an[g] = arow[4 * (sn + g)]; }
#pragma unroll
            for (int g = 0; g < G; ++g)
            {
                if (((s + g) & 1) == 0) { acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[g], b[s + g], acc0, 0, 0, 0); }
                else { acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[g], b[s + g], acc1, 0, 0, 0); }
            }
#pragma unroll
            for (int g = 0; g < G; ++g) { a[g] = an[g]; }
        }
    }
    return acc0 + acc1;
}

// TWO row tiles of one mel tile at once (plans with a mel tile per wave: more than 64 mel bins): the weights are fetched once
// for both, every output is ONE accumulator filled in k order -- the reference's `sm += Yf[f]*w` chain exactly
// (c/kaldi_fbank.c:337-345), where one row tile alone needs two alternating accumulators to hide the MFMA's latency -- and the
// two independent chains hide it instead.  (Before: a wave ran (tile, row tile 0) and then (tile + 4, row tile 1), with the
// second job's weights requested when the first was done, i.e. an L2 latency in the middle of the phase.)
template <int KP, int G>
__device__ __forceinline__ void mfma_tile_pre_rows2(const float *arow0, const float *arow1, const float (&b)[KP], const int ksteps, f32x4 &o0, f32x4 &o1)
{
    f32x4 acc0 = {0.0f, 0.0f, 0.0f, 0.0f}, acc1 = {0.0f, 0.0f, 0.0f, 0.0f};
    if (ksteps > 0)
    {
        float a0[G], a1[G];
#pragma unroll
        for (int g = 0; g < G; ++g) { a0[g] = arow0[4 * g]; a1[g] = arow1[4 * g]; }
#pragma unroll
        for (int s = 0; s < KP; s += G)
        {
            if (s < ksteps)
            {
                const int sn = min(s + G, ksteps - G);                      // the last group reloads itself (harmless)
                float a0n[G], a1n[G];
#pragma unroll
                for (int g = 0; g < G; ++g) { a0n[g] = arow0[4 * (sn + g)]; a1n[g] = arow1[4 * (sn + g)]; }
#pragma unroll
                for (int g = 0; g < G; ++g)
                {
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[g], b[s + g], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[g], b[s + g], acc1, 0, 0, 0);
                }
#pragma unroll
                for (int g = 0; g < G; ++g) { a0[g] = a0n[g]; a1[g] = a1n[g]; }
            }
        }
    }
    o0 = acc0; o1 = acc1;
}

// fbank / mfcc / plp.  KIND = SPEECH_FBANK | SPEECH_MFCC | SPEECH_PLP;  P = LPC order (plp only, compile-time so that
// Levinson-Durbin and the cepstral recursion stay in registers).
// PAIR (256-point plans): 8 frames per wave as 4 pairs (frame_power_pair), 64 frames per tile, a 64 x 130
// power tile; NZ then counts rows of 16 samples.
// WIDE (1024-point plans): 2 frames per wave, 32 lanes each (frame_power_wide), 16 frames per tile, a 16 x 514 power
// tile; NZ counts rows of 64 samples.
// XW (2048-point plans): 1 frame per wave (frame_power_xwide), 8 frames per tile, an 8 x 1026 power tile that fills half
// of a 16-row matrix-core tile (rows 8..15 repeat rows 0..7 and are dropped); NZ counts rows of 128 samples.  Only the twiddles
// are staged in LDS (the window is read through L1), so that at 48 kHz / 25 ms three workgroups still share a CU.
template <int NZ, int KIND, int P, bool DITHER, bool PAIR, bool WIDE = false, bool XW = false>
__global__ __launch_bounds__(NTM, (XW && NZ > 10) ? 4 : 6) void fast_mel_kernel(const FastParams p)      // 6 waves per SIMD = 3 workgroups per CU: at most 80 VGPRs (2048-point frames above 1280 samples: LDS holds two workgroups, 128 VGPRs)
{
    static_assert(!(PAIR && WIDE) && !(XW && (PAIR || WIDE)), "one mode at a time");
    constexpr int FWK = XW ? 1 : (WIDE ? 2 : (PAIR ? 2 * QF : QF));         // frames per wave
    constexpr int TILEK = MW * FWK;                                         // frames per workgroup
    constexpr int RTK = XW ? 1 : TILEK / 16;                                // 16-frame row tiles
    constexpr int PPK = XW ? PPX : (WIDE ? PPW : (PAIR ? PP2 : PP));        // pitch of the power tile
    // plp runs in TWO launches: this kernel stops at the autocorrelations r[0..12] (written to the frame's output row, which has
    // C >= 13 columns), plp_tail_kernel below turns every row into cepstra in place, one lane per frame over dense waves.
    // (Rounds 1-2 ran Levinson-Durbin inside this kernel -- one lane per frame of a 32-frame tile, later of four sub-tiles =
    // 128 frames walked in a loop; workgroups that loop run their phases in step and lose what staggered one-tile workgroups
    // gain from each other: plp cost mfcc + 14-16 %.)
    constexpr int TABK = XW ? TABX_TWF : (WIDE ? TABW_FLOATS : TAB_FLOATS);        // floats of tables in front of the slices (2048-point plans: the twiddles; the window stays in memory)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x;
    int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    // Everything the start-up needs from the argument segment is read in ONE batch and held (the compiler had spread these
    // reads over the start-up as eight s_load - s_waitcnt round trips in a row, in front of the two descriptor reads that do
    // depend on each other)
    const float *a_tab = p.tab, *a_mat2 = p.mat2, *a_X = p.X;
    const short *a_X16 = p.X16;
    uint32_t a_ntiles = p.ntiles;                                          // (not the descriptor arrays' addresses: a read through a held pointer is no longer a scalar load)
    int a_m2 = p.m2, a_ws = p.ws, a_stp = p.stp, a_L = p.L, a_snip = p.snip_edges;
    // (not volatile: a volatile statement counts as a possible store, and the descriptor reads below would stop being scalar loads)
    asm("" : "+s"(a_tab), "+s"(a_mat2), "+s"(a_X), "+s"(a_X16), "+s"(a_ntiles), "+s"(a_m2), "+s"(a_ws), "+s"(a_stp), "+s"(a_L), "+s"(a_snip));
    const TileDesc tile = p.tiles[xcd_tile(blockIdx.x, a_ntiles)];
    float *s_tab = smem;
    float *s_m2 = smem + TABK;                                              // mat2, when staged (a_m2 floats)
    float *s_sl = s_m2 + a_m2;                                              // the waves' slices
    float *s_aud = s_sl + wv * a_ws;                                 // the wave's slice: audio, then transposition image
    float *s_tr = s_aud;
    // after the FFTs the slices are dead and hold the TILEK x PPK power tile
    float *s_pow = s_sl;
    // private region behind the slices: frame energies | fbank: output tile | mfcc, plp: mel tile.  The mfcc / plp output
    // tile (and plp's autocorrelations) are written after the power tile is dead and reuse its place.
    float *s_e = s_sl + MW * a_ws;
    float *s_out = (KIND == SPEECH_FBANK) ? s_e + TILEK : s_pow;
    float *s_mel = s_e + TILEK;
    // tables (and the staged matrix) straight into LDS, 1 KB per wave-instruction (buffer_load_dwordx4 ... lds), the chunks dealt
    // to the waves: nothing waits for them before the barrier in front of the FFT.  (As float4 loads through registers every
    // thread stood in an s_waitcnt vmcnt(0) -- an L2 round trip -- before the tile descriptor was even requested.)
    {
        const __amdgpu_buffer_rsrc_t rt = __builtin_amdgcn_make_buffer_rsrc((void *)a_tab, 0, TABK * 4, BUF_FLAGS);
        for (int c = wv; 256 * c < TABK; c += MW)
        {
            if (256 * c + 4 * lane < TABK) { __builtin_amdgcn_raw_ptr_buffer_load_lds(rt, (__attribute__((address_space(3))) void *)(s_tab + 256 * c), 16, lane * 16, 1024 * c, 0, 0); }
        }
        const __amdgpu_buffer_rsrc_t rm = __builtin_amdgcn_make_buffer_rsrc((void *)a_mat2, 0, a_m2 * 4, BUF_FLAGS);
        for (int c = MW - 1 - wv; 256 * c < a_m2; c += MW)
        {
            if (256 * c + 4 * lane < a_m2) { __builtin_amdgcn_raw_ptr_buffer_load_lds(rm, (__attribute__((address_space(3))) void *)(s_m2 + 256 * c), 16, lane * 16, 1024 * c, 0, 0); }
        }
    }
    const UttDesc utt = p.utts[tile.utt];
    // Kernel arguments and descriptor fields that the phases BEHIND the FFT use are read here and kept in scalar registers
    // (pinned below, once the audio loads are on their way).  Left to itself the compiler re-reads each of them from the
    // argument segment where it is used -- cheap in instructions, but every such s_load is followed by an s_waitcnt, and in
    // the contraction and output phases, which are chains of short latency-bound steps between workgroup barriers, that
    // was 22 scalar-cache round trips in a row (fbank-80: mel_kbeg / mel_ksteps after the barrier, mel_cnt, then mel_first
    // and D, then D again, lg, Y ...) on the critical path of a workgroup's life.
    int a_K2 = p.K2, a_ntm = p.ntile_m, a_ntc = p.ntile_c, a_amp = p.amp, a_D = p.D, a_ue = p.use_energy, a_MP = p.MP;
    int a_B = p.B, a_lg = p.lg, a_Cp = p.Cp, a_C = p.C, a_cm = p.compress_mode;
    const float *a_melw = p.melw;
    const bool rows2 = (RTK == 2) && (KIND != SPEECH_PLP) && (a_ntm == MW);
    // the wave's first mel job: tile t0 (plans with a mel tile per wave: both row tiles of tile wv, see below)
    const int t0 = rows2 ? wv : ((wv < RTK * a_ntm) ? wv / RTK : 0);
    int m_ks = p.mel_ksteps[t0], m_kbeg = p.mel_kbeg[t0], m_first = p.mel_first[t0], m_cnt = p.mel_cnt[t0], m_woff = p.mel_woff[t0];
    // the tile's nframes x D outputs are one contiguous run of Y
    float *y = p.Y + utt.y_off + (unsigned long long)tile.frame0 * (unsigned long long)a_D;
    int nout = (int)tile.nframes * a_D;
    const int fq = wv * FWK;                                                // the wave's first frame within the sub-tile
    const int nfr = min((int)tile.nframes - fq, FWK);
    if (nfr > 0)
    {
        const long long w0 = (long long)tile.frame0 + fq;
        const long long s0 = w0 * a_stp + (a_snip ? 0 : (long long)(a_stp / 2) - (long long)(a_L / 2));   // c/kaldi_mfcc.c:81-82
        if (!(SPEECH_ABL(p) & 2)) load_span<XW>(s_aud, a_X + ((SPEECH_ABL(p) & 16) ? 0 : utt.x_off), a_X16 ? a_X16 + utt.x_off : nullptr, s0, (nfr - 1) * a_stp + a_L, (long long)utt.n, lane, (SPEECH_ABL(p) & 256) ? 0 : (a_ws & ~255));
    }
    asm volatile("" : "+s"(a_K2), "+s"(a_ntm), "+s"(a_ntc), "+s"(a_ws), "+s"(a_amp), "+s"(a_D), "+s"(a_ue), "+s"(a_MP), "+s"(a_B), "+s"(a_lg),
                      "+s"(a_Cp), "+s"(a_m2), "+s"(a_C), "+s"(a_cm), "+s"(a_mat2), "+s"(a_melw), "+s"(m_ks), "+s"(m_kbeg), "+s"(m_first), "+s"(m_cnt),
                      "+s"(m_woff), "+s"(y), "+s"(nout));
    __syncthreads();
    const float *s_tab_l = s_tab;
    constexpr int fbase = 0;

    const int f = lane >> 4, k2 = lane & 15;
    float pa[9], pb[9], rawe = 1.0f;
    f2 rawe2 = {1.0f, 1.0f};
    if (nfr > 0 && XW)
    {
        const unsigned long long dkey = DITHER ? dither_frame_key(mix64(p.seed ^ ((unsigned long long)tile.uid << 32)),
                                                                  (unsigned long long)((long long)tile.frame0 + fbase + fq)) : 0ull;
        frame_power_xwide<NZ, true, DITHER, true>(s_aud, s_tab_l, s_tr, lane, p, dkey, pa, pb, rawe);
    }
    else if (nfr > 0 && WIDE)
    {
        // lanes 32 j .. 32 j + 31: frame fq + j (a missing second frame computes on whatever the slice holds; its row is never stored)
        const unsigned long long dkey = DITHER ? dither_frame_key(mix64(p.seed ^ ((unsigned long long)tile.uid << 32)),
                                                                  (unsigned long long)((long long)tile.frame0 + fbase + fq + (lane >> 5))) : 0ull;
        frame_power_wide<NZ, true, DITHER>(s_aud + (lane >> 5) * p.stp, s_tab_l, s_tr, lane, p, dkey, pa, pb, rawe);
    }
    else if (nfr > 0 && PAIR)
    {
        // lanes 16 f .. 16 f + 15: frames fq + 2 f (A) and fq + 2 f + 1 (B; a missing B repeats A, its row is never stored)
        const unsigned long long ku = DITHER ? mix64(p.seed ^ ((unsigned long long)tile.uid << 32)) : 0ull;
        const unsigned long long wa = (unsigned long long)((long long)tile.frame0 + fbase + fq + 2 * f);
        const unsigned long long ka = DITHER ? dither_frame_key(ku, wa) : 0ull, kb = DITHER ? dither_frame_key(ku, wa + 1) : 0ull;
        frame_power_pair<NZ, true, DITHER>(s_aud + 2 * f * p.stp, (2 * f + 1 < nfr) ? p.stp : 0, s_tab_l, s_tr, lane, p, ka, kb, pa, pb, rawe2);
    }
    else if (nfr > 0)
    {
        const unsigned long long dkey = DITHER ? dither_frame_key(mix64(p.seed ^ ((unsigned long long)tile.uid << 32)),
                                                                  (unsigned long long)((long long)tile.frame0 + fbase + fq + f)) : 0ull;
        frame_power<NZ, true, DITHER>(s_aud + f * p.stp, s_tab_l, s_tr, lane, p, dkey, pa, pb, rawe);
    }
    else
    {
#pragma unroll
        for (int i = 0; i < 9; ++i) { pa[i] = 0.0f; pb[i] = 0.0f; }
    }
    // weights of the wave's first mel job and of its first DCT / IDFT job: requested now, used after the barriers below
    // (what the contractions derive from the lane number and from the plan's sizes must not be computed at the top of the
    // kernel and carried through the FFT -- that spills; hence the opaque copies)
    int lane_m = lane, ks2 = a_K2 >> 2;                                     // ks2: k-steps of the second contraction
    asm volatile("" : "+v"(lane_m), "+s"(ks2));
    const int fr0 = (lane_m >> 4) * 4, col = lane_m & 15;
    const int njob_m = RTK * a_ntm;
    const int njob_2 = (KIND == SPEECH_MFCC) ? RTK * a_ntc : ((KIND == SPEECH_PLP) ? RTK : 0);
    // plans with a mel tile per wave (ntile_m == MW, i.e. more than 64 mel bins) and two row tiles (rows2): wave w contracts BOTH
    // row tiles of mel tile w with one set of weights (mfma_tile_pre_rows2); otherwise job j = (mel tile j / RTK, row tile j % RTK)
    // (not plp: more than 64 mel bins is not its use, and the second code path cost its kernel 5 spilled registers)
    float wpre[KPRE];
    {
        const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void *)(a_melw + m_woff), 0, (wv < njob_m) ? 256 * m_ks : 0, BUF_FLAGS);
        weights_issue(wpre, rb, lane_m * 4, 256);
    }
    // plp: the equal-loudness factor of the lane's mel bin in the wave's first job travels with the weights (read in the job's
    // epilogue it was a dependent L2 access per job with nothing to hide behind: the mel phase of plp cost 1.66 x mfcc's)
    float eqv = 0.0f;
    if (KIND == SPEECH_PLP && wv < njob_m) { if (col < m_cnt) { eqv = p.aux[m_first + col]; } }

    // The power tile takes the slices' place.  When a slice is exactly as long as the wave's rows of the tile (the usual
    // case: ws = FWK * PPK), a wave's rows lie in its own slice and it only has to be done with its own transposition image.
    if (a_ws != FWK * PPK) { __syncthreads(); } else { wave_sync(); }
    if (SPEECH_ABL(p) & 4) { if (pa[0] + pb[3] + rawe + rawe2.x == 123.456f) { p.Y[tid] = pa[1]; } return; }      // diagnostics: front end + FFT only

    // power (c/kaldi_mfcc.c:307-310; amplitude if amp, c/kaldi_fbank.c:312-316) -> TILEK x 257 tile, column 257 zero
    if (XW)
    {
        // group 0: bins k, 1024 - k; group 1: 512 - k, 512 + k; group 2: 256 - k, 768 + k; group 3: 256 + k, 768 - k
        // (k = k2 + 16 k1; frame_power_xwide names the repeats that are not written); column 1025 is the zero a padded k-loop may touch
        float *pr = s_pow + fq * PPK;
        const int gg = lane >> 4;
        if (a_amp)
        {
#pragma unroll
            for (int k1 = 0; k1 < 9; ++k1) { pa[k1] = __fsqrt_rn(pa[k1]); pb[k1] = __fsqrt_rn(pb[k1]); }
        }
        const int ia = (gg == 0) ? 0 : ((gg == 1) ? 512 : 256), ib = (gg == 0) ? 1024 : ((gg == 1) ? 512 : 768);
        const int sa = (gg == 0 || gg == 3) ? 1 : -1, sb = (gg == 1 || gg == 2) ? 1 : -1;
#pragma unroll
        for (int k1 = 0; k1 < 8; ++k1)
        {
            const int k = k2 + 16 * k1;
            const bool rep_a = (k == 0 && gg == 3), rep_b = (k == 0 && (gg & 1));
            if (!rep_a) { pr[ia + sa * k] = pa[k1]; }
            if (!rep_b) { pr[ib + sb * k] = pb[k1]; }
        }
        if (k2 == 0 && gg < 2) { pr[ia + sa * 128] = pa[8]; pr[ib + sb * 128] = pb[8]; }
        if (lane == 0) { pr[1025] = 0.0f; s_e[fbase + fq] = rawe; }
    }
    else if (WIDE)
    {
        // even-group lanes hold bins k and 512 - k, odd-group lanes 256 - k and 256 + k (k = k2 + 16 k1); column 513 is the zero
        // a padded k-loop may touch
        float *pr = s_pow + (fq + (lane >> 5)) * PPK;
        const int hh = (lane >> 4) & 1;
        if (a_amp)
        {
#pragma unroll
            for (int k1 = 0; k1 < 9; ++k1) { pa[k1] = __fsqrt_rn(pa[k1]); pb[k1] = __fsqrt_rn(pb[k1]); }
        }
#pragma unroll
        for (int k1 = 0; k1 < 8; ++k1)
        {
            const int k = k2 + 16 * k1;
            pr[hh ? 256 - k : k] = pa[k1];
            pr[hh ? 256 + k : 512 - k] = pb[k1];
        }
        if (k2 == 0 && hh == 0) { pr[128] = pa[8]; pr[384] = pb[8]; pr[513] = 0.0f; s_e[fbase + fq + (lane >> 5)] = rawe; }
    }
    else if (PAIR)
    {
        // pa[k1] / pb[k1]: bin k2 + 16 k1 of frames A / B, 129 bins each; column 129 is the zero a padded k-loop may touch
        float *pra = s_pow + (fq + 2 * f) * PPK, *prb = pra + PPK;
        if (a_amp)
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
            s_e[fbase + fq + 2 * f] = rawe2.x; s_e[fbase + fq + 2 * f + 1] = rawe2.y;
        }
    }
    else
    {
        float *pr = s_pow + (fq + f) * PPK;
        if (a_amp)
        {
#pragma unroll
            for (int k1 = 0; k1 < 9; ++k1) { pa[k1] = __fsqrt_rn(pa[k1]); pb[k1] = __fsqrt_rn(pb[k1]); }
        }
        // bins 256 - k2 - 16 k1 through ONE base register and immediate offsets (written as pr[NC - k2 - 16 * k1] the compiler
        // rebuilt the address for every k1: 14 vector instructions per quad)
        int kb = NC - 112 - k2;
        asm volatile("" : "+v"(kb));
        float *prb = pr + kb;
#pragma unroll
        for (int k1 = 0; k1 < 8; ++k1)
        {
            pr[k2 + 16 * k1] = pa[k1];
            prb[16 * (7 - k1)] = pb[k1];
        }
        if (k2 == 0) { pr[128] = pa[8]; pr[NC + 1] = 0.0f; s_e[fbase + fq + f] = rawe; }       // column 257: the zero a padded k-loop may touch
    }
    __syncthreads();
    if (SPEECH_ABL(p) & 32) { if (s_pow[tid] == 123.456f) { p.Y[tid] = 1.0f; } return; }

    // mel filterbank: frames x bins . bins x mels, banded (c/kaldi_mfcc.c:313-320, c/kaldi_fbank.c:337-345)
    const int eo = (KIND == SPEECH_FBANK && a_ue) ? 1 : 0;
    if (RTK == 2 && KIND != SPEECH_PLP && rows2)
    {
        const int t = wv;
        const int ks = m_ks;
        const float *arow = s_pow + col * PPK + m_kbeg + (lane_m >> 4);
        f32x4 acc[2];
        mfma_tile_pre_rows2<KPRE, 4>(arow, arow + 16 * PPK, wpre, ks, acc[0], acc[1]);
        if (ks > KPRE)                                                      // a tile of few, wide filters: the rest of its k-steps in the loop form
        {
            const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void *)(a_melw + m_woff), 0, 256 * ks, BUF_FLAGS);
            acc[0] += mfma_tile4(arow + 4 * KPRE, rb, lane_m * 4 + 256 * KPRE, ks - KPRE);
            acc[1] += mfma_tile4(arow + 16 * PPK + 4 * KPRE, rb, lane_m * 4 + 256 * KPRE, ks - KPRE);
        }
        const int m = m_first + col;
        const bool live = col < m_cnt;
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
        {
#pragma unroll
            for (int r = 0; r < 4; ++r)
            {
                const int fr = 16 * rt + fr0 + r;
                const float v = acc[rt][r];
                if (KIND == SPEECH_FBANK)
                {
                    if (live) { s_out[fr * a_D + eo + m] = v; }                                 // c/kaldi_fbank.c:345
                }
                else if (KIND == SPEECH_MFCC)
                {
                    if (live) { s_mel[fr * a_MP + m] = floor_log(v); }                          // c/kaldi_mfcc.c:318
                }
                else if (live)
                {
                    const float x = v * eqv;                                                    // c/kaldi_plp.c:339-358
                    const float c = (a_cm == 1) ? x : ((a_cm == 2) ? __fsqrt_rn(x) : (PLP_POW_SPLIT ? pow_split(x, p.compress) : __builtin_amdgcn_exp2f(p.compress * __log2f(x))));
                    s_mel[fr * a_MP + 1 + m] = c;
                    if (m == 0) { s_mel[fr * a_MP] = c; }
                    if (m == a_B - 1) { for (int j = a_B + 1; j < a_K2; ++j) { s_mel[fr * a_MP + j] = (j == a_B + 1) ? c : 0.0f; } }
                }
            }
        }
    }
    else
    for (int job = wv; job < njob_m; job += MW)
    {
        const int t = job / RTK, rt = job - t * RTK;                          // mel tile, 16-frame row tile; its numbers are in m_* (the first job's since the top of the kernel, a later job's since the job before)
        const int ks = m_ks;
        const float *arow = s_pow + (16 * rt + (XW ? (col & 7) : col)) * PPK + m_kbeg + (lane_m >> 4);
        f32x4 acc = mfma_tile_pre<KPRE, 4>(arow, wpre, ks);
        if (ks > KPRE)                                                      // a tile of few, wide filters: the rest of its k-steps in the loop form
        {
            const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void *)(a_melw + m_woff), 0, 256 * ks, BUF_FLAGS);
            acc += mfma_tile4(arow + 4 * KPRE, rb, lane_m * 4 + 256 * KPRE, ks - KPRE);
        }
        const int m = m_first + col;                                        // this lane's mel bin; the tile holds mel_cnt[t] <= 16 of them
        const bool live_c = col < m_cnt;
        const float eq = eqv;                                               // plp: this job's equal-loudness factor; the next job's is requested now
        if (job + MW < njob_m)                                              // the wave's next job: its numbers, and its weights travel during this one's epilogue
        {
            const int tn = (job + MW) / RTK;
            m_ks = p.mel_ksteps[tn]; m_kbeg = p.mel_kbeg[tn]; m_first = p.mel_first[tn]; m_cnt = p.mel_cnt[tn]; m_woff = p.mel_woff[tn];
            const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void *)(a_melw + m_woff), 0, 256 * m_ks, BUF_FLAGS);
            weights_issue(wpre, rb, lane_m * 4, 256);
            if (KIND == SPEECH_PLP) { eqv = (col < m_cnt) ? p.aux[m_first + col] : 0.0f; }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r)
        {
            const int fr = 16 * rt + fr0 + r;
            const float v = acc[r];
            const bool live = live_c && (!XW || fr < TILEK);
            if (KIND == SPEECH_FBANK)
            {
                if (live) { s_out[fr * a_D + eo + m] = v; }                                 // c/kaldi_fbank.c:345 (last-bin quirk: duplicated matrix column); the log is taken on the way out
            }
            else if (KIND == SPEECH_MFCC)
            {
                if (live) { s_mel[fr * a_MP + m] = floor_log(v); }                          // c/kaldi_mfcc.c:318
            }
            else
            {
                // equal loudness, compression, end duplication (c/kaldi_plp.c:339-358)
                if (live)
                {
                    const float x = v * eq;
                    const float c = (a_cm == 1) ? x : ((a_cm == 2) ? __fsqrt_rn(x) : (PLP_POW_SPLIT ? pow_split(x, p.compress) : __builtin_amdgcn_exp2f(p.compress * __log2f(x))));
                    s_mel[fr * a_MP + 1 + m] = c;
                    if (m == 0) { s_mel[fr * a_MP] = c; }
                    if (m == a_B - 1) { for (int j = a_B + 1; j < a_K2; ++j) { s_mel[fr * a_MP + j] = (j == a_B + 1) ? c : 0.0f; } }
                }
            }
        }
    }
    if (KIND == SPEECH_FBANK && eo && tid < TILEK) { s_out[tid * a_D] = a_lg ? s_e[tid] : pos_log(s_e[tid]); }     // c/kaldi_fbank.c:299 (its log, too, on the way out when lg)
    if (KIND == SPEECH_MFCC && tid < 4 * TILEK) { const int j = a_B + (tid & 3); if (j < a_K2) { s_mel[(tid >> 2) * a_MP + j] = 0.0f; } }      // k-padding of the DCT operand
    __syncthreads();
    if (SPEECH_ABL(p) & 64) { if (s_mel[tid] == 123.456f) { p.Y[tid] = 1.0f; } return; }

    if (KIND == SPEECH_MFCC)
    {
        // DCT-II x lifter as a second contraction (c/kaldi_mfcc.c:323-333); the last-cepstrum quirk is a duplicated column
        for (int job = wv; job < njob_2; job += MW)
        {
            const int t = job / RTK, rt = job - t * RTK;
            const float *arow = s_mel + (16 * rt + (XW ? (col & 7) : col)) * a_MP + (lane_m >> 4);
            const int boff = (lane_m >> 4) * a_Cp + 16 * t + col;
            const f32x4 acc = a_m2 ? mfma_tile(arow, s_m2 + boff, 4 * a_Cp, ks2) : mfma_tile(arow, a_mat2 + boff, 4 * a_Cp, ks2);
            const int c = 16 * t + col;
#pragma unroll
            for (int r = 0; r < 4; ++r)
            {
                const int fr = 16 * rt + fr0 + r;
                // c0 := log energy (c/kaldi_mfcc.c:327); with C == 2 the last-cepstrum copy (:333) then takes the energy too
                const bool en = a_ue && (c == 0 || (c == 1 && a_C == 2));
                if (c < a_C && (!XW || fr < TILEK)) { s_out[fr * a_C + c] = en ? pos_log(s_e[fr]) : acc[r]; }
            }
        }
        __syncthreads();
    }
    if (KIND == SPEECH_PLP)
    {
        // autocorrelation = IDFT bases x compressed spectrum, one MFMA tile per 16 frames.  The
        // power tile is dead since the barrier above: r[0..12] of frame fr goes to columns 0..12 of its output row (C >= 13;
        // plp_tail_kernel rewrites the whole row), the frame's energy -- when it is used -- to the batch's energy scratch
        if (wv < RTK)
        {
            const float *arow = s_mel + (16 * wv + (XW ? (col & 7) : col)) * a_MP + (lane_m >> 4);
            const f32x4 acc = a_m2 ? mfma_tile(arow, s_m2 + lane_m, 64, ks2) : mfma_tile(arow, a_mat2 + lane_m, 64, ks2);
#pragma unroll
            for (int r = 0; r < 4; ++r) { if (col <= P && (!XW || fr0 + r < TILEK)) { s_out[(16 * wv + fr0 + r) * a_C + col] = acc[r]; } }
        }
        if (a_ue && tid < (int)tile.nframes)
        {
            // row number of the utterance's first frame in the energy scratch: y_off / C (exact in double for any offset below 2^53)
            unsigned long long q = (unsigned long long)((double)utt.y_off / (double)a_C);
            if (q * (unsigned long long)a_C > utt.y_off) { --q; } else if ((q + 1ull) * (unsigned long long)a_C <= utt.y_off) { ++q; }
            p.plp_en[q + tile.frame0 + (unsigned)tid] = s_e[tid];
        }
        __syncthreads();
    }

    // 16 bytes per lane when the run allows it (D a multiple of 4 and the run's first float on a 16-byte boundary, in memory and
    // in LDS): 1 KB of whole cache lines per wave-instruction, a quarter of the loop trips and of their address arithmetic
    // (fbank-80: 2 trips of a descriptor store instead of 5 of load - wait - log - 64-bit address - store)
    const bool vec4 = ((a_D & 3) == 0) && (((unsigned long long)y & 15ull) == 0ull) && ((((unsigned)(s_out - smem)) & 3u) == 0u);
    if (vec4)
    {
        const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc((void *)y, 0, nout * 4, BUF_FLAGS);
        const int n4 = nout >> 2;
        for (int i = tid; i < n4; i += NTM)
        {
            f32x4 v = reinterpret_cast<const f32x4 *>(s_out)[i];
            if (KIND == SPEECH_FBANK && a_lg) { v = f32x4{floor_log(v.x), floor_log(v.y), floor_log(v.z), floor_log(v.w)}; }
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), ry, i * 16, 0, 0);
        }
    }
    else if (KIND == SPEECH_FBANK && a_lg)
    {
        // log-mel: the log is taken here, on the dense output tile (every lane a live value), not in the contraction's
        // epilogue, where 16-column tiles of 10 mels leave a third of the lanes computing logs of padding
        for (int i = tid; i < nout; i += NTM) { y[i] = floor_log(s_out[i]); }
    }
    else
    {
        for (int i = tid; i < nout; i += NTM) { y[i] = s_out[i]; }
    }
}

template <int NZ, bool DITHER, bool PAIR, bool WIDE = false, bool XW = false>
kernel_fn mel_for_kind(int kind)
{
    switch (kind)
    {
        case SPEECH_FBANK: return fast_mel_kernel<NZ, SPEECH_FBANK, 0, DITHER, PAIR, WIDE, XW>;
        case SPEECH_MFCC: return fast_mel_kernel<NZ, SPEECH_MFCC, 0, DITHER, PAIR, WIDE, XW>;
        default: return fast_mel_kernel<NZ, SPEECH_PLP, 12, DITHER, PAIR, WIDE, XW>;
    }
}

#if defined(SPEECH_MEL_XWIDE_TU)
// the 2048-point instantiations are a translation unit of their own (fast_mel_xwide.hip), built beside this one
template <bool DITHER>
kernel_fn mel_xwide_for_nz(int nz, int kind)
{
    switch (nz)
    {
        case 9: return mel_for_kind<9, DITHER, false, false, true>(kind);
        case 10: return mel_for_kind<10, DITHER, false, false, true>(kind);
        case 11: return mel_for_kind<11, DITHER, false, false, true>(kind);
        case 12: return mel_for_kind<12, DITHER, false, false, true>(kind);
        case 13: return mel_for_kind<13, DITHER, false, false, true>(kind);
        case 14: return mel_for_kind<14, DITHER, false, false, true>(kind);
        case 15: return mel_for_kind<15, DITHER, false, false, true>(kind);
        default: return mel_for_kind<16, DITHER, false, false, true>(kind);
    }
}

}  // namespace

kernel_fn fast_mel_xwide_kernel_for(int nz, int kind, bool dither) { return dither ? mel_xwide_for_nz<true>(nz, kind) : mel_xwide_for_nz<false>(nz, kind); }

#else
// plp, second launch: every frame's row holds r[0..12] (columns 0..P of its C >= P + 1 columns); one lane per frame turns it
// into the cepstra in place -- Levinson-Durbin with Kaldi's 1 - k^2 >= 1e-5 guard, LPC -> cepstrum, lifter, in the oracle's
// sequential order (reference intent c/kaldi_plp.c:206,380-403).  The recursions are ~500
// dependent instructions per frame whatever the lane count, so one lane per frame over full waves is the issue-optimal shape;
// a workgroup takes TAIL_FRAMES consecutive frames of one utterance, whose rows are one contiguous run of Y: they come in and
// go out as whole cache lines through LDS (element i of the run lives at i + i / 32: a lane walks its row at stride C, and
// the padding keeps that conflict-free for even C as well).
constexpr int TAIL_FRAMES = 128;

// (Two frames per lane in packed-f32 registers halve the instruction count and change nothing -- 30.1 against 28.9 us for
// 1 000 x 10 s: the launch moves 2 x 52 MB in that time, i.e. it costs what its traffic costs.)
template <int P>
__global__ __launch_bounds__(TAIL_FRAMES) void plp_tail_kernel(const PlpTailParams q)
{
    extern __shared__ __attribute__((aligned(16))) float s_rows[];
    const UttDesc utt = q.utts[blockIdx.x];
    const long long W = (long long)utt.W;
    const long long row0 = (long long)blockIdx.y * TAIL_FRAMES;
    if (row0 >= W) { return; }
    const int tid = threadIdx.x, C = q.C;
    const int nr = (int)min((long long)TAIL_FRAMES, W - row0);
    float *y = q.Y + utt.y_off + (unsigned long long)row0 * (unsigned long long)C;
    const int n = nr * C;
    {
        // the run's floats, sixteen loads per lane in flight before the first is parked (a loop of load - wait - store pairs
        // pays one memory latency per 128 floats: 13 in a row for C = 13); loads past the run return 0 (range check)
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)y, 0, n * 4, BUF_FLAGS);
        for (int i0 = 0; i0 < n; i0 += 16 * TAIL_FRAMES)
        {
            float v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) { v[u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (i0 + tid) * 4, u * TAIL_FRAMES * 4, 0)); }
#pragma unroll
            for (int u = 0; u < 16; ++u) { const int i = i0 + u * TAIL_FRAMES + tid; if (i < n) { s_rows[i + (i >> 5)] = v[u]; } }
        }
    }
    __syncthreads();
    if (tid < nr)
    {
        const int base = tid * C;
        float a[P], tmp[P], cep[P], r[P + 1];
#pragma unroll
        for (int i = 0; i <= P; ++i) { r[i] = s_rows[base + i + ((base + i) >> 5)]; }
        float E = r[0];
#pragma unroll
        for (int i = 0; i < P; ++i)
        {
            float ki = r[i + 1];
#pragma unroll
            for (int j = 0; j < i; ++j) { ki += a[j] * r[i - j]; }
            ki = PLP_IEEE_DIV ? ki / E : ki * fast_rcp(E);              // the oracle's (and Kaldi's) k /= E
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
            cep[i] = -a[i] - sum * (1.0f / (float)(i + 1));
        }
        const float *lift = q.aux + q.B;
        float prev = 0.0f;
#pragma unroll
        for (int k = 0; k <= P; ++k)
        {
            if (k < C)
            {
                float v = (k == 0) ? floor_log(E) : cep[k > 0 ? k - 1 : 0];
                v *= lift[k];
                if (k == 0 && q.use_energy)                                                           // c/kaldi_plp.c:397
                {
                    unsigned long long e0 = (unsigned long long)((double)utt.y_off / (double)C);
                    if (e0 * (unsigned long long)C > utt.y_off) { --e0; } else if ((e0 + 1ull) * (unsigned long long)C <= utt.y_off) { ++e0; }
                    v = pos_log(q.en[e0 + (unsigned long long)row0 + (unsigned)tid]);
                }
                if (k == C - 1 && C >= 2) { v = prev; }                                               // c/kaldi_plp.c:403
                s_rows[base + k + ((base + k) >> 5)] = v; prev = v;
            }
        }
        for (int k = P + 1; k < C; ++k) { const float v = (k == C - 1) ? prev : 0.0f; s_rows[base + k + ((base + k) >> 5)] = v; prev = v; }
    }
    __syncthreads();
    for (int i = tid; i < n; i += TAIL_FRAMES) { y[i] = s_rows[i + (i >> 5)]; }
}

}  // namespace

hipError_t launch_plp_tail(const PlpTailParams &q, uint32_t nutt, uint32_t max_frames, hipStream_t s)
{
    if (nutt == 0 || max_frames == 0) { return hipSuccess; }
    const size_t n = (size_t)TAIL_FRAMES * (size_t)q.C;
    const size_t lds = (n + n / 32u + 1u) * sizeof(float);                  // C <= 96: at most 50.7 KB, inside the 64 KB a launch may ask for without opting in
    hipLaunchKernelGGL(plp_tail_kernel<12>, dim3(nutt, (max_frames + TAIL_FRAMES - 1) / TAIL_FRAMES), dim3(TAIL_FRAMES), lds, s, q);
    return hipGetLastError();
}

namespace {

template <bool DITHER>
kernel_fn mel_wide_for_nz(int nz, int kind)
{
    switch (nz)
    {
        case 9: return mel_for_kind<9, DITHER, false, true>(kind);
        case 10: return mel_for_kind<10, DITHER, false, true>(kind);
        case 11: return mel_for_kind<11, DITHER, false, true>(kind);
        case 12: return mel_for_kind<12, DITHER, false, true>(kind);
        case 13: return mel_for_kind<13, DITHER, false, true>(kind);
        case 14: return mel_for_kind<14, DITHER, false, true>(kind);
        case 15: return mel_for_kind<15, DITHER, false, true>(kind);
        default: return mel_for_kind<16, DITHER, false, true>(kind);
    }
}

// dithered frames have their own instantiations: the generator's temporaries would otherwise cost the plain kernels
// 13-24 VGPRs and a resident workgroup per CU
template <bool PAIR>
kernel_fn mel_for_nz(int nz, int kind, bool dither)
{
    switch (nz)
    {
        case 9: return dither ? mel_for_kind<9, true, PAIR>(kind) : mel_for_kind<9, false, PAIR>(kind);
        case 10: return dither ? mel_for_kind<10, true, PAIR>(kind) : mel_for_kind<10, false, PAIR>(kind);
        case 11: return dither ? mel_for_kind<11, true, PAIR>(kind) : mel_for_kind<11, false, PAIR>(kind);
        case 12: return dither ? mel_for_kind<12, true, PAIR>(kind) : mel_for_kind<12, false, PAIR>(kind);
        case 13: return dither ? mel_for_kind<13, true, PAIR>(kind) : mel_for_kind<13, false, PAIR>(kind);
        case 14: return dither ? mel_for_kind<14, true, PAIR>(kind) : mel_for_kind<14, false, PAIR>(kind);
        case 15: return dither ? mel_for_kind<15, true, PAIR>(kind) : mel_for_kind<15, false, PAIR>(kind);
        default: return dither ? mel_for_kind<16, true, PAIR>(kind) : mel_for_kind<16, false, PAIR>(kind);
    }
}

}  // namespace

kernel_fn fast_mel_kernel_for(int nz, int kind, bool dither, bool pair, bool wide, bool xwide)
{
    if (xwide) { return fast_mel_xwide_kernel_for(nz, kind, dither); }
    if (wide) { return dither ? mel_wide_for_nz<true>(nz, kind) : mel_wide_for_nz<false>(nz, kind); }
    return pair ? mel_for_nz<true>(nz, kind, dither) : mel_for_nz<false>(nz, kind, dither);
}
#endif

}  // namespace speech
