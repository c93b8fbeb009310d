// fast_spectrogram.hip -- the tuned spectrogram kernel (see fast_common.hpp for the per-frame front end + FFT).
#include "fast_common.hpp"

namespace speech {
namespace {

// PAIR (256-point plans): a wave carries 8 frames as 4 pairs (frame_power_pair), a sub-tile is 32 frames, a
// tile 128; NZ then counts rows of 16 samples.
// WIDE (1024-point plans): 2 frames per wave, 32 lanes each (frame_power_wide), a sub-tile is 8 frames, a tile 32;
// NZ counts rows of 64 samples.
// XW (2048-point plans): 1 frame per wave on all 64 lanes (frame_power_xwide), a sub-tile is 4 frames, a tile 16; NZ counts
// rows of 128 samples.
template <int NZ, bool DITHER, bool PCM, bool PAIR, bool WIDE = false, bool XW = false>
__global__ __launch_bounds__(NT, 4) void fast_spectrogram_kernel(const FastParams p)
{
    static_assert(!(PAIR && WIDE) && !(XW && (PAIR || WIDE)), "one mode at a time");
    constexpr int TABK = XW ? TABX_FLOATS : (WIDE ? TABW_FLOATS : TAB_FLOATS);
    constexpr int NPK = (WIDE || XW) ? 20 : NPRE;                           // prefetch registers: 2 frames of a 1024-point plan span up to 1280 samples at 25 / 10 ms (a 2048-point frame of more than 1280 samples is loaded when its turn comes)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    float *s_tab = smem;
    float *s_aud = smem + TABK + wv * p.ws;                                 // the wave's slice: audio, then transposition image
    float *s_tr = s_aud;
    for (int i = tid; i < TABK / 4; i += NT) { reinterpret_cast<float4 *>(s_tab)[i] = reinterpret_cast<const float4 *>(p.tab)[i]; }

    const TileDesc tile = p.tiles[xcd_tile(blockIdx.x, p.ntiles)];
    const UttDesc utt = p.utts[tile.utt];
    const float *xu = p.X + ((SPEECH_ABL(p) & 16) ? 0 : utt.x_off);
    const short *xu16 = p.X16 ? p.X16 + utt.x_off : nullptr;
    const long long N = (long long)utt.n;
    constexpr int FW = XW ? 1 : (WIDE ? 2 : (PAIR ? 2 * QF : QF));          // frames per wave
    constexpr int SUBF = (NT / 64) * FW;                                    // frames per sub-tile
    const int nsub = ((int)tile.nframes + SUBF - 1) / SUBF;
    const unsigned long long dkey_utt = DITHER ? mix64(p.seed ^ ((unsigned long long)tile.uid << 32)) : 0ull;
    long long s0; int need;
    int nfr = quad_geom(p, tile, wv * FW, s0, need, FW);
    if (nfr > 0) { load_span<XW>(s_aud, xu, xu16, s0, need, N, lane); }
    __syncthreads();                                                        // tables (and, wave-locally, the first span) are in LDS

    const int f = lane >> 4, k2 = lane & 15;
    for (int it = 0; it < nsub; ++it)
    {
        const int fq = it * SUBF + wv * FW;                                  // this wave's quad in sub-tile `it`
        long long s0n; int needn;
        const int nfrn = quad_geom(p, tile, fq + SUBF, s0n, needn, FW);
        Prefetch<NPK> pf;
        prefetch_issue<PCM, NPK>(pf, p, xu, xu16, s0n, needn, N, nfrn, lane);
        float pa[9], pb[9], rawe = 1.0f;
        f2 rawe2 = {1.0f, 1.0f};
        if (nfr > 0)
        {
            if (XW)
            {
                const unsigned long long dkey = DITHER ? dither_frame_key(dkey_utt, (unsigned long long)((long long)tile.frame0 + fq)) : 0ull;
                frame_power_xwide<NZ, false, DITHER>(s_aud, s_tab + opaque_zero(), s_tr, lane, p, dkey, pa, pb, rawe);
            }
            else if (WIDE)
            {
                // lanes 32 j .. 32 j + 31: frame fq + j (a missing second frame computes on whatever the slice holds and is dropped)
                const unsigned long long dkey = DITHER ? dither_frame_key(dkey_utt, (unsigned long long)((long long)tile.frame0 + fq + (lane >> 5))) : 0ull;
                frame_power_wide<NZ, false, DITHER>(s_aud + (lane >> 5) * p.stp, s_tab + opaque_zero(), s_tr, lane, p, dkey, pa, pb, rawe);
            }
            else if (PAIR)
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
        prefetch_park<PCM, NPK, XW>(pf, s_aud, xu, xu16, s0n, needn, N, nfrn, lane, SPEECH_ABL(p));
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
            if (XW)
            {
                // group 0: bins k, 1024 - k; group 1: 512 - k, 512 + k; group 2: 256 - k, 768 + k; group 3: 256 + k, 768 - k; D = 1025
                float *sw = s_aud + p.span;
                const int gg = lane >> 4;
                const int ia = (gg == 0) ? 0 : ((gg == 1) ? 512 : 256), ib = (gg == 0) ? 1024 : ((gg == 1) ? 512 : 768);
                const int sa = (gg == 0 || gg == 3) ? 1 : -1, sb = (gg == 1 || gg == 2) ? 1 : -1;
#pragma unroll
                for (int k1 = 0; k1 < 8; ++k1)
                {
                    const int k = k2 + 16 * k1;
                    const f2 l = spec_log2((k == 0 && gg == 0) ? 4.0f * rawe : pa[k1], pb[k1]);
                    if (!(k == 0 && gg == 3)) { sw[ia + sa * k] = l.x; }
                    if (!(k == 0 && (gg & 1)) && !(k == 0 && gg == 0)) { sw[ib + sb * k] = l.y; }
                    if (gg == 0 && k == 1) { sw[1024] = l.y; }                                  // last bin := the one before (c/kaldi_spectrogram.c:245-247)
                }
                if (k2 == 0 && gg < 2) { const f2 l = spec_log2(pa[8], pb[8]); sw[ia + sa * 128] = l.x; sw[ib + sb * 128] = l.y; }
            }
            else if (WIDE)
            {
                // even-group lanes hold bins k and 512 - k, odd-group lanes 256 - k and 256 + k (k = k2 + 16 k1); D = 513
                float *sw = s_aud + p.span + (lane >> 5) * D;
                const int hh = (lane >> 4) & 1;
#pragma unroll
                for (int k1 = 0; k1 < 8; ++k1)
                {
                    const int k = k2 + 16 * k1;
                    const f2 l = spec_log2((k == 0 && hh == 0) ? 4.0f * rawe : pa[k1], pb[k1]);
                    sw[hh ? 256 - k : k] = l.x;
                    if (hh) { sw[256 + k] = l.y; }
                    else if (k > 0) { sw[512 - k] = l.y; }
                    if (hh == 0 && k == 1) { sw[512] = l.y; }                                   // last bin := the one before (c/kaldi_spectrogram.c:245-247)
                }
                if (k2 == 0 && hh == 0) { const f2 l = spec_log2(pa[8], pb[8]); sw[128] = l.x; sw[384] = l.y; }
            }
            else if (PAIR)
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
                (void *)(p.Y + ((SPEECH_ABL(p) & 8) ? (unsigned long long)((blockIdx.x & 31u) * TILE_SPEC + (fq & (TILE_SPEC - 1))) * D      /* diagnostics */
                                               : utt.y_off + (unsigned long long)((long long)tile.frame0 + fq) * D)), 0, (SPEECH_ABL(p) & 1) ? 0 : nfr * D * 4, BUF_FLAGS);
            if (nfr == FW)
            {
                // FW * D floats = FW D / 4 chunks of 16 bytes (4 x 257: four full instructions and one lane; 8 x 129: 258 chunks)
                for (int c = lane; c < FW * D / 4; c += 64)
                {
                    const u32x4 v = *reinterpret_cast<const u32x4 *>(so + 4 * c);
                    __builtin_amdgcn_raw_buffer_store_b128(v, ry, c * 16, 0, 0);
                }
                if ((WIDE || XW) && lane < (WIDE ? 2 : 1)) { const int i = 4 * (FW * D / 4) + lane; __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, so[i]), ry, i * 4, 0, 0); }      // 2 x 513 = 256 chunks + 2
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

template <int NZ, bool PAIR>
kernel_fn spectrogram_for(bool dither, bool pcm)
{
    // dithered frames and 16-bit PCM input have their own instantiations: the generator's temporaries would cost the plain
    // kernel its registers, and a run-time choice of the load instruction a wait right after the prefetch (fast_common.hpp)
    if (dither) { return pcm ? fast_spectrogram_kernel<NZ, true, true, PAIR> : fast_spectrogram_kernel<NZ, true, false, PAIR>; }
    return pcm ? fast_spectrogram_kernel<NZ, false, true, PAIR> : fast_spectrogram_kernel<NZ, false, false, PAIR>;
}

template <bool PAIR>
kernel_fn spectrogram_for_nz(int nz, bool dither, bool pcm)
{
    switch (nz)
    {
        case 9: return spectrogram_for<9, PAIR>(dither, pcm);
        case 10: return spectrogram_for<10, PAIR>(dither, pcm);
        case 11: return spectrogram_for<11, PAIR>(dither, pcm);
        case 12: return spectrogram_for<12, PAIR>(dither, pcm);
        case 13: return spectrogram_for<13, PAIR>(dither, pcm);
        case 14: return spectrogram_for<14, PAIR>(dither, pcm);
        case 15: return spectrogram_for<15, PAIR>(dither, pcm);
        default: return spectrogram_for<16, PAIR>(dither, pcm);
    }
}


template <int NZ>
kernel_fn spectrogram_wide_for(bool dither, bool pcm)
{
    if (dither) { return pcm ? fast_spectrogram_kernel<NZ, true, true, false, true> : fast_spectrogram_kernel<NZ, true, false, false, true>; }
    return pcm ? fast_spectrogram_kernel<NZ, false, true, false, true> : fast_spectrogram_kernel<NZ, false, false, false, true>;
}

template <int NZ>
kernel_fn spectrogram_xwide_for(bool dither, bool pcm)
{
    if (dither) { return pcm ? fast_spectrogram_kernel<NZ, true, true, false, false, true> : fast_spectrogram_kernel<NZ, true, false, false, false, true>; }
    return pcm ? fast_spectrogram_kernel<NZ, false, true, false, false, true> : fast_spectrogram_kernel<NZ, false, false, false, false, true>;
}

kernel_fn spectrogram_xwide_for_nz(int nz, bool dither, bool pcm)
{
    switch (nz)
    {
        case 9: return spectrogram_xwide_for<9>(dither, pcm);
        case 10: return spectrogram_xwide_for<10>(dither, pcm);
        case 11: return spectrogram_xwide_for<11>(dither, pcm);
        case 12: return spectrogram_xwide_for<12>(dither, pcm);
        case 13: return spectrogram_xwide_for<13>(dither, pcm);
        case 14: return spectrogram_xwide_for<14>(dither, pcm);
        case 15: return spectrogram_xwide_for<15>(dither, pcm);
        default: return spectrogram_xwide_for<16>(dither, pcm);
    }
}

kernel_fn spectrogram_wide_for_nz(int nz, bool dither, bool pcm)
{
    switch (nz)
    {
        case 9: return spectrogram_wide_for<9>(dither, pcm);
        case 10: return spectrogram_wide_for<10>(dither, pcm);
        case 11: return spectrogram_wide_for<11>(dither, pcm);
        case 12: return spectrogram_wide_for<12>(dither, pcm);
        case 13: return spectrogram_wide_for<13>(dither, pcm);
        case 14: return spectrogram_wide_for<14>(dither, pcm);
        case 15: return spectrogram_wide_for<15>(dither, pcm);
        default: return spectrogram_wide_for<16>(dither, pcm);
    }
}

}  // namespace

kernel_fn fast_spectrogram_kernel_for(int nz, bool dither, bool pcm, bool pair, bool wide, bool xwide)
{
    if (xwide) { return spectrogram_xwide_for_nz(nz, dither, pcm); }
    if (wide) { return spectrogram_wide_for_nz(nz, dither, pcm); }
    return pair ? spectrogram_for_nz<true>(nz, dither, pcm) : spectrogram_for_nz<false>(nz, dither, pcm);
}

}  // namespace speech
