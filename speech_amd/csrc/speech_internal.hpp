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
    uint32_t pad;
};

// Everything a per-frame kernel needs besides the data (device pointers are to the plan's tables).
template <typename T>
struct FrameParams {
    const T *X;
    const short *X16;            // non-null: 16-bit PCM input (same element offsets), X is then unused
    T *Y;
    const UttDesc *utts;
    const TileDesc *tiles;
    uint32_t ntiles;
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
    T *mel_out;                  // optional scratch [frames x B] (plp two-kernel path), else nullptr
};

struct MeanNormParams {
    void *Y;
    const UttDesc *utts;
    uint32_t nutt;
    int D;
};

// launchers (defined in the kernel translation units)
hipError_t launch_generic_f32(const FrameParams<float> &p, hipStream_t s);
hipError_t launch_generic_f64(const FrameParams<double> &p, hipStream_t s);
size_t generic_lds_bytes(int nfft, int B, int C, int P, size_t elem);
hipError_t launch_mean_norm_f32(const MeanNormParams &p, hipStream_t s);
hipError_t launch_mean_norm_f64(const MeanNormParams &p, hipStream_t s);

constexpr int GENERIC_TILE_FRAMES = 8;
constexpr int GENERIC_MAX_NFFT = 8192;

}  // namespace speech
