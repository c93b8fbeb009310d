// speech_api.hip -- host layer of libspeech.so: option validation, per-plan tables, batch geometry,
// launches, and the ten drop-in entry points of include/codee_speech.h.
//
// Tables are built on the host with the same libm calls and in the same precision as the
// reference builds them per call (window c/kaldi_mfcc.c:105-135, mel c/kaldi_mfcc.c:157-188,
// lifter c/kaldi_mfcc.c:200-210, equal loudness c/kaldi_plp.c:183-193), then kept on the device.
// All arithmetic on audio happens in HIP kernels; there is no CPU fallback.
#include "speech_internal.hpp"
#include "../../include/codee_speech.h"
#include <cfloat>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <atomic>
#include <memory>
#include <mutex>
#include <new>
#include <type_traits>

#ifndef M_PI
#define M_PI 3.141592653589793238462643383279502884
#endif

namespace speech {

// fast-path hooks (fast_kernels.hip)
struct FastTables;
bool fast_supported(const speech_opts &o, size_t L, size_t nfft);
int fast_tile_frames(const speech_opts &o, size_t L, size_t nfft);
FastTables *fast_tables_create(const speech_opts &o, size_t L, size_t stp, size_t nfft, size_t B, size_t C, size_t P,
                               const std::vector<float> &win, const std::vector<float> &f2b, const std::vector<int> &beg,
                               const std::vector<int> &num, const std::vector<float> &dct, const std::vector<float> &lift,
                               const std::vector<float> &eql, const std::vector<float> &idft, std::string *err);
void fast_tables_destroy(FastTables *t);
hipError_t launch_fast(const FastTables *t, const FrameParams<float> &p, hipStream_t s, hipEvent_t ev0, hipEvent_t ev1);

static thread_local std::string g_err;

int fail(const char *fn, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = std::string("error in ") + fn + ": " + buf;
    fprintf(stderr, "%s\n", g_err.c_str());
    return 1;
}

static const char *kind_name(int kind, int dtype)
{
    static const char *names[4][2] = {{"kaldi_spectrogram_s", "kaldi_spectrogram_d"}, {"kaldi_fbank_s", "kaldi_fbank_d"},
                                      {"kaldi_mfcc_s", "kaldi_mfcc_d"}, {"kaldi_plp_s", "kaldi_plp_d"}};
    if (kind < 0 || kind > 3) { return "speech"; }
    return names[kind][dtype == SPEECH_F64 ? 1 : 0];
}

struct Sizes { size_t L, stp, nfft, F, D; int log2n; };

// Scalar validation and size derivation in the working precision: c/kaldi_mfcc.c:54-78 (and the same prologue
// in c/kaldi_spectrogram.c:48-66, c/kaldi_fbank.c:61-82, c/kaldi_plp.c:55-79).
template <typename T>
static int validate(const speech_opts &o, Sizes *sz)
{
    const char *fn = kind_name(o.kind, o.dtype);
    const T eps = std::is_same<T, float>::value ? (T)FLT_EPSILON : (T)DBL_EPSILON;
    if (o.kind < 0 || o.kind > 3) { return fail("speech_plan_create", "unknown kind %d", o.kind); }
    const T sr = (T)o.sr, fl = (T)o.frame_length, fs = (T)o.frame_shift;
    if (!(sr >= eps)) { return fail(fn, "sr must be positive"); }
    if (!(fl >= eps)) { return fail(fn, "frame_length must be positive"); }
    if (!(fs >= eps)) { return fail(fn, "frame_shift must be positive"); }
    if ((T)o.dither < (T)0) { return fail(fn, "dither must be nonnegative"); }
    if ((T)o.preemph < (T)0) { return fail(fn, "preemph coeff must be nonnegative"); }
    if (o.kind != SPEECH_SPECTROGRAM)
    {
        if ((T)o.lof < (T)0) { return fail(fn, "lof (low-freq cutoff) must be nonnegative"); }
        if (!((T)o.hif > (T)o.lof)) { return fail(fn, "hif (high-freq cutoff) must be > lof"); }
        if (o.B < 1u) { return fail(fn, "B (num mel bins) must be positive"); }
    }
    if (o.kind == SPEECH_MFCC || o.kind == SPEECH_PLP)
    {
        if (o.C < 1u) { return fail(fn, "C (num cepstral coeffs) must be positive"); }
        if (o.C > o.B) { return fail(fn, "C (num cepstral coeffs) must be <= B (num mel bins)"); }
        if ((T)o.Q < (T)0) { return fail(fn, "Q (lifter coeff) must be nonnegative"); }
    }
    const T Lf = sr * fl / (T)1000.0, Sf = sr * fs / (T)1000.0;
    if (!(Lf < (T)2147483647.0) || !(Sf < (T)2147483647.0)) { return fail(fn, "frame_length/frame_shift too large (int-indexed, c/kaldi_mfcc.c:82-84)"); }
    const size_t L = (size_t)Lf;
    if (L < 1u) { return fail(fn, "frame_length must be greater than 1 sample"); }
    const size_t stp = (size_t)Sf;
    if (stp < 1u) { return fail(fn, "frame_shift must be greater than 1 sample"); }
    if (L < 3u) { return fail(fn, "frame_length must be at least 3 samples"); }     // defined divergence, DESIGN.md "edge cases"
    size_t nfft = 1u; int lg2 = 0;
    while (nfft < L) { nfft *= 2u; ++lg2; }
    if (nfft > (size_t)GENERIC_MAX_NFFT) { return fail(fn, "frame_length of %zu samples needs a %zu-point FFT; this build supports up to %d", L, nfft, GENERIC_MAX_NFFT); }
    if (o.B > 4096u) { return fail(fn, "B (num mel bins) above 4096 is not supported"); }
    sz->L = L; sz->stp = stp; sz->nfft = nfft; sz->F = nfft / 2u + 1u; sz->log2n = lg2;
    if (o.kind == SPEECH_SPECTROGRAM) { sz->D = sz->F; }
    else if (o.kind == SPEECH_FBANK) { sz->D = o.B + (o.use_energy ? 1u : 0u); }
    else { sz->D = o.C; }
    return 0;
}

static int window_known(const char *w)
{
    return strncmp(w, "rectangular", 11u) == 0 || strncmp(w, "blackman", 8u) == 0 || strncmp(w, "hamming", 7u) == 0 ||
           strncmp(w, "hann", 4u) == 0 || strncmp(w, "povey", 5u) == 0;
}

template <typename T> struct HM;
template <> struct HM<float> {
    static float cos(float x) { return cosf(x); } static float sin(float x) { return sinf(x); }
    static float log(float x) { return logf(x); } static float exp(float x) { return expf(x); }
    static float pow(float x, float y) { return powf(x, y); } static float sqrt(float x) { return sqrtf(x); }
};
template <> struct HM<double> {
    static double cos(double x) { return ::cos(x); } static double sin(double x) { return ::sin(x); }
    static double log(double x) { return ::log(x); } static double exp(double x) { return ::exp(x); }
    static double pow(double x, double y) { return ::pow(x, y); } static double sqrt(double x) { return ::sqrt(x); }
};

template <typename T>
struct HostTables {
    std::vector<T> win, twr, twi, f2b, dct, lift, eql, idft;
    std::vector<int> beg, num, off;
};

template <typename T>
static void build_tables(const speech_opts &o, const Sizes &sz, size_t P, HostTables<T> *t)
{
    const size_t L = sz.L, nfft = sz.nfft, F = sz.F, B = o.B, C = o.C;
    // window: c/kaldi_mfcc.c:105-135 (prefix match in this order)
    t->win.resize(L);
    const char *wt = o.win_type;
    if (strncmp(wt, "rectangular", 11u) == 0) { for (size_t l = 0; l < L; ++l) { t->win[l] = (T)1; } }
    else
    {
        const T a = std::is_same<T, float>::value ? (T)(float)(2.0 * M_PI / (double)(L - 1u)) : (T)(2.0 * M_PI / (double)(L - 1u));
        if (strncmp(wt, "blackman", 8u) == 0)
        {
            const T coeff = (T)0.42, coeff1 = (T)1 - coeff;       // sic: c/kaldi_mfcc.c:116
            for (size_t l = 0; l < L; ++l) { t->win[l] = coeff - (T)0.5 * HM<T>::cos((T)l * a) + coeff1 * HM<T>::cos((T)(2u * l) * a); }
        }
        else if (strncmp(wt, "hamming", 7u) == 0) { for (size_t l = 0; l < L; ++l) { t->win[l] = (T)0.54 - (T)0.46 * HM<T>::cos((T)l * a); } }
        else if (strncmp(wt, "hann", 4u) == 0) { for (size_t l = 0; l < L; ++l) { t->win[l] = (T)0.5 - (T)0.5 * HM<T>::cos((T)l * a); } }
        else { for (size_t l = 0; l < L; ++l) { t->win[l] = HM<T>::pow((T)0.5 - (T)0.5 * HM<T>::cos((T)l * a), (T)0.85); } }
    }
    // twiddles of the generic radix-2 FFT
    t->twr.resize(nfft / 2u); t->twi.resize(nfft / 2u);
    for (size_t k = 0; k < nfft / 2u; ++k)
    {
        const long double ang = -2.0L * 3.141592653589793238462643383279502884L * (long double)k / (long double)nfft;
        t->twr[k] = (T)cosl(ang); t->twi[k] = (T)sinl(ang);
    }
    if (o.kind == SPEECH_SPECTROGRAM) { return; }

    // sparse mel table: c/kaldi_mfcc.c:157-188
    const T sr = (T)o.sr, lof = (T)o.lof, hif = (T)o.hif;
    const T finc = sr / (T)nfft;
    const T lomel = (T)1127.0 * HM<T>::log((T)1 + lof / (T)700.0);
    const T himel = (T)1127.0 * HM<T>::log((T)1 + hif / (T)700.0);
    const T dmel = (himel - lomel) / (T)(B + 1u);
    std::vector<T> mels(F);
    for (size_t f = 0; f < F; ++f) { mels[f] = (T)1127.0 * HM<T>::log((T)1 + (T)f * finc / (T)700.0); }
    T lmel = lomel, cmel = lmel + dmel, rmel = cmel + dmel;
    t->beg.resize(B); t->num.resize(B); t->off.resize(B);
    for (size_t b = B; b > 0u; --b)
    {
        const size_t i = B - b;
        size_t f = 0u;
        while (f < F && mels[f] < lmel) { ++f; }
        t->beg[i] = (int)f; t->off[i] = (int)t->f2b.size();
        while (f < F && mels[f] < cmel) { t->f2b.push_back((mels[f] - lmel) / dmel); ++f; }
        while (f < F && mels[f] < rmel) { t->f2b.push_back((rmel - mels[f]) / dmel); ++f; }
        t->num[i] = (int)f - t->beg[i];
        lmel = cmel; cmel = rmel;
        rmel = (b == 2u) ? himel : rmel + dmel;
    }
    if (t->f2b.empty()) { t->f2b.push_back((T)0); }

    if (o.kind == SPEECH_MFCC)
    {
        // lifter with folded DCT scaling: c/kaldi_mfcc.c:200-210
        t->lift.resize(C);
        const T Q = (T)o.Q;
        const T sc = (T)1 / HM<T>::sqrt((T)(2u * B));
        if (Q > (T)FLT_EPSILON) { for (size_t c = 0; c < C; ++c) { t->lift[c] = (T)1 + (T)0.5 * Q * HM<T>::sin((T)M_PI * (T)c / Q); } }
        else { for (size_t c = 0; c < C; ++c) { t->lift[c] = (T)1; } }
        for (size_t c = 1; c < C; ++c) { t->lift[c] *= sc; }
        t->lift[0] *= (T)0.5 / HM<T>::sqrt((T)B);
        // FFTW REDFT10 as a matrix (plan at c/kaldi_mfcc.c:195): Yd[k] = 2 sum_n Xd[n] cos(pi (n+1/2) k / B)
        t->dct.resize(C * B);
        for (size_t c = 0; c < C; ++c) { for (size_t b = 0; b < B; ++b) {
            t->dct[c * B + b] = (T)(2.0L * cosl(3.141592653589793238462643383279502884L * ((long double)b + 0.5L) * (long double)c / (long double)B)); } }
    }
    if (o.kind == SPEECH_PLP)
    {
        // equal loudness: c/kaldi_plp.c:183-193
        t->eql.resize(B);
        T cm = lomel + dmel;
        for (size_t b = 0; b < B; ++b, cm += dmel)
        {
            const T cf = (T)700.0 * (HM<T>::exp(cm / (T)1127.0) - (T)1);
            const T fsq = cf * cf;
            const T fsub = fsq / (fsq + (T)160000.0);
            t->eql[b] = fsub * fsub * ((fsq + (T)1440000.0) / (fsq + (T)9610000.0));
        }
        // Kaldi idft bases
        t->idft.resize((P + 1u) * (B + 2u));
        const T angle = (T)M_PI / (T)(B + 1u);
        const T scale = (T)1 / ((T)2 * (T)(B + 1u));
        for (size_t i = 0; i <= P; ++i)
        {
            t->idft[i * (B + 2u)] = scale;
            for (size_t j = 1; j <= B; ++j) { t->idft[i * (B + 2u) + j] = (T)2 * scale * HM<T>::cos(angle * (T)i * (T)j); }
            t->idft[i * (B + 2u) + B + 1u] = scale * HM<T>::cos(angle * (T)i * (T)(B + 1u));
        }
        t->lift.resize(C);
        const T Q = (T)o.Q, cs = (T)o.cep_scale;
        if (Q > (T)FLT_EPSILON) { for (size_t c = 0; c < C; ++c) { t->lift[c] = ((T)1 + (T)0.5 * Q * HM<T>::sin((T)M_PI * (T)c / Q)) * cs; } }
        else { for (size_t c = 0; c < C; ++c) { t->lift[c] = cs; } }
    }
}

}  // namespace speech

using namespace speech;

struct speech_plan {
    speech_opts o;
    int device;
    Sizes sz;
    size_t P;
    bool tuned;
    bool quad64;                 // double precision on quad_f64.hip rather than the generic kernel
    std::vector<void *> dev_allocs;
    void *d_win, *d_twr, *d_twi, *d_f2b, *d_dct, *d_lift, *d_eql, *d_idft;
    int *d_beg, *d_num, *d_off;
    FastTables *fast;
};

struct speech_batch {
    speech_plan *plan;
    std::vector<UttDesc> utts;
    std::vector<TileDesc> tiles;
    UttDesc *d_utts;
    TileDesc *d_tiles;
    double *d_mom;               // post-processing: [nutt][2][D] moments, then [2][D] scale / offset (allocated on first use)
    unsigned long long *d_zrow;  // splice + subsample: first output row of each utterance for (zrow_n, zrow_off); allocated on first use
    std::vector<unsigned long long> zrow;
    int zrow_n, zrow_off;
    int yoff_pow2;               // largest of 1, 2, 4 that divides every utterance's output offset (-1: not computed yet)
    unsigned long long *d_ids;   // speech_batch_synth_audio: the utterances' corpus numbers (allocated on first use)
    unsigned long long max_n;    // longest utterance, samples
    unsigned long long max_w;    // longest utterance, frames
    float *d_plp_en;             // tuned plp with use_energy: frame energies between the plan's two launches, one float per output row
    size_t plp_rows;             // rows that scratch must hold: max over utterances of y_off / C + W (0: not needed)
    size_t total_samples, total_out, total_frames;
    hipEvent_t ev0, ev1;
};

namespace {

#define HIP_TRY(fn, expr)                                                                         \
    do {                                                                                          \
        hipError_t e_ = (expr);                                                                   \
        if (e_ != hipSuccess) { return fail(fn, "HIP call failed: %s (%s)", #expr, hipGetErrorString(e_)); } \
    } while (0)

struct DeviceGuard {
    int prev; bool ok;
    explicit DeviceGuard(int dev) : prev(-1), ok(false)
    {
        if (hipGetDevice(&prev) != hipSuccess) { prev = -1; }
        ok = (hipSetDevice(dev) == hipSuccess);
    }
    ~DeviceGuard() { if (prev >= 0) { (void)hipSetDevice(prev); } }
};

template <typename V>
int upload(speech_plan *p, const std::vector<V> &h, void **d)
{
    *d = nullptr;
    if (h.empty()) { return 0; }
    HIP_TRY("speech_plan_create", hipMalloc(d, h.size() * sizeof(V)));
    p->dev_allocs.push_back(*d);
    HIP_TRY("speech_plan_create", hipMemcpy(*d, h.data(), h.size() * sizeof(V), hipMemcpyHostToDevice));
    return 0;
}

template <typename T>
int plan_build(speech_plan *p)
{
    HostTables<T> t;
    build_tables<T>(p->o, p->sz, p->P, &t);
    if (upload(p, t.win, &p->d_win) || upload(p, t.twr, &p->d_twr) || upload(p, t.twi, &p->d_twi) || upload(p, t.f2b, &p->d_f2b) ||
        upload(p, t.dct, &p->d_dct) || upload(p, t.lift, &p->d_lift) || upload(p, t.eql, &p->d_eql) || upload(p, t.idft, &p->d_idft) ||
        upload(p, t.beg, (void **)&p->d_beg) || upload(p, t.num, (void **)&p->d_num) || upload(p, t.off, (void **)&p->d_off)) { return 1; }
    p->quad64 = (sizeof(T) == 8) && quad_f64_supported(p->o, p->sz.nfft, p->P);
    return 0;
}

template <>
int plan_build<float>(speech_plan *p)
{
    HostTables<float> t;
    build_tables<float>(p->o, p->sz, p->P, &t);
    if (upload(p, t.win, &p->d_win) || upload(p, t.twr, &p->d_twr) || upload(p, t.twi, &p->d_twi) || upload(p, t.f2b, &p->d_f2b) ||
        upload(p, t.dct, &p->d_dct) || upload(p, t.lift, &p->d_lift) || upload(p, t.eql, &p->d_eql) || upload(p, t.idft, &p->d_idft) ||
        upload(p, t.beg, (void **)&p->d_beg) || upload(p, t.num, (void **)&p->d_num) || upload(p, t.off, (void **)&p->d_off)) { return 1; }
    p->tuned = false; p->fast = nullptr;
    if (!p->o.force_generic && fast_supported(p->o, p->sz.L, p->sz.nfft))
    {
        std::string err;
        p->fast = fast_tables_create(p->o, p->sz.L, p->sz.stp, p->sz.nfft, p->o.B, p->o.C, p->P, t.win, t.f2b, t.beg, t.num, t.dct, t.lift, t.eql, t.idft, &err);
        if (!p->fast && !err.empty()) { return fail("speech_plan_create", "tuned-kernel tables: %s", err.c_str()); }
        p->tuned = (p->fast != nullptr);                 // declined without a message: this option set runs on the generic kernels
    }
    return 0;
}

template <typename T>
void fill_params(const speech_batch *b, const void *X, const short *X16, void *Y, FrameParams<T> *fp)
{
    const speech_plan *p = b->plan;
    const speech_opts &o = p->o;
    memset(fp, 0, sizeof(*fp));
    fp->X = (const T *)X; fp->X16 = X16; fp->Y = (T *)Y;
    fp->utts = b->d_utts; fp->tiles = b->d_tiles; fp->ntiles = (uint32_t)b->tiles.size();
    fp->nutt = (uint32_t)b->utts.size(); fp->max_frames = (uint32_t)b->max_w; fp->plp_en = (T *)b->d_plp_en;
    fp->kind = o.kind;
    fp->L = (int)p->sz.L; fp->stp = (int)p->sz.stp; fp->nfft = (int)p->sz.nfft; fp->F = (int)p->sz.F; fp->log2n = p->sz.log2n;
    fp->D = (int)p->sz.D;
    fp->snip_edges = o.snip_edges ? 1 : 0; fp->dc0 = o.dc0 ? 1 : 0; fp->raw_energy = o.raw_energy ? 1 : 0;
    fp->use_energy = o.use_energy ? 1 : 0; fp->amp = (o.kind == SPEECH_FBANK && o.amp) ? 1 : 0; fp->lg = o.lg ? 1 : 0;   // amp, lg: fbank only (c/kaldi_fbank.c:312-316)
    const bool spec = (o.kind == SPEECH_SPECTROGRAM);
    fp->want_pre = spec ? (o.raw_energy ? 1 : 0) : ((o.use_energy && o.raw_energy) ? 1 : 0);     // c/kaldi_spectrogram.c:203 vs c/kaldi_mfcc.c:274
    fp->want_post = spec ? (o.raw_energy ? 0 : 1) : ((o.use_energy && !o.raw_energy) ? 1 : 0);   // c/kaldi_spectrogram.c:224 vs c/kaldi_mfcc.c:295
    fp->pre_before_dc = (o.dtype == SPEECH_F64 && o.kind == SPEECH_MFCC) ? 1 : 0;                // c/kaldi_mfcc.c:576-591
    fp->preemph = (T)o.preemph; fp->dither = (T)o.dither;
    fp->do_preemph = !(p->sz.L < 2u || (T)o.preemph < (T)FLT_EPSILON);                           // c/kaldi_mfcc.c:282
    fp->do_dither = ((T)o.dither > (T)FLT_EPSILON);                                              // c/kaldi_mfcc.c:243
    fp->seed = o.dither_seed;
    fp->win = (const T *)p->d_win; fp->tw_re = (const T *)p->d_twr; fp->tw_im = (const T *)p->d_twi;
    fp->B = spec ? 0 : (int)o.B;
    fp->f2b = (const T *)p->d_f2b; fp->mel_beg = p->d_beg; fp->mel_num = p->d_num; fp->mel_off = p->d_off;
    fp->C = (o.kind == SPEECH_MFCC || o.kind == SPEECH_PLP) ? (int)o.C : 0;
    fp->dct = (const T *)p->d_dct; fp->lift = (const T *)p->d_lift;
    fp->P = (o.kind == SPEECH_PLP) ? (int)p->P : 0;
    fp->compress = (T)o.compress;
    fp->compress_mode = ((T)o.compress == (T)1) ? 1 : (((T)o.compress == (T)0.5) ? 2 : 0);       // c/kaldi_plp.c:341,347
    fp->eql = (const T *)p->d_eql; fp->idft = (const T *)p->d_idft;
}

// shared by mn0 and the post-processing entry points
int post_begin(const char *fn, const speech_batch *b, PostParams *pp, uint32_t *max_frames)
{
    if (!b || !b->plan) { return fail(fn, "null batch"); }
    memset(pp, 0, sizeof(*pp));
    pp->utts = b->d_utts; pp->nutt = (uint32_t)b->utts.size(); pp->D = (int)b->plan->sz.D; pp->rows_per_wg = 128;
    unsigned long long mw = 0;
    for (const UttDesc &u : b->utts) { mw = std::max(mw, u.W); }
    *max_frames = (uint32_t)mw;
    return 0;
}

// splice: can the rows be copied in 16-byte units?
int splice_vec16(speech_batch *b, const void *Y, const void *Z)
{
    if (b->yoff_pow2 < 0)
    {
        unsigned long long m = 0;
        for (const UttDesc &u : b->utts) { m |= u.y_off; }
        b->yoff_pow2 = (m & 1u) ? 1 : ((m & 2u) ? 2 : 4);
    }
    const size_t vw = b->plan->o.dtype == SPEECH_F64 ? 2u : 4u;
    return (b->plan->sz.D % vw == 0u && (size_t)b->yoff_pow2 >= vw && ((uintptr_t)Y & 15u) == 0u && ((uintptr_t)Z & 15u) == 0u) ? 1 : 0;
}

// the batch's moment buffer: [nutt][2][D] per-utterance moments | [2][D] scale / offset of speech_batch_affine | the levels
// of speech_batch_stats' reduction tree (ceil(nutt / 64) rows, then ceil of that / 64, ...)
size_t mom_doubles(size_t nutt, size_t D) { return (nutt + 1u + nutt / (POST_REDUCE_GROUP - 1) + 8u) * 2u * D; }

int post_moments(const char *fn, speech_batch *b, const void *Y, PostParams *pp, hipStream_t s)
{
    if (!b->d_mom) { HIP_TRY(fn, hipMalloc((void **)&b->d_mom, mom_doubles(pp->nutt, pp->D) * sizeof(double))); }
    pp->Yin = Y; pp->mom = b->d_mom;
    HIP_TRY(fn, launch_post(POST_MOMENTS, b->plan->o.dtype == SPEECH_F64, *pp, 0, s));
    return 0;
}


// ---- static device scratch of speech_batch_cmvn_noalloc (include/speech_hip.h) ----
// Per-utterance moments [nutt][2][D] in double, for callers that cannot allocate.  Each GPU holds its own copy of the array
// at its own address: the address is asked for once per device (speech_plan_create, with that device current) and kept;
// host code uses only that pointer, never the array's name.
constexpr size_t SCRATCH_DOUBLES = (size_t)512 * 1024;                      // 4 MiB per GPU, held until the process ends
static __device__ __align__(256) double scratch[SCRATCH_DOUBLES];
constexpr int SCRATCH_MAX_DEVICES = 64;
std::atomic<double *> g_scratch_ptr[SCRATCH_MAX_DEVICES];
std::mutex g_scratch_mu;

// with device `dev` current
int scratch_init(const char *fn, int dev)
{
    if (dev < 0 || dev >= SCRATCH_MAX_DEVICES) { return 0; }                // such a device simply has no no-alloc entry point
    if (g_scratch_ptr[dev].load() != nullptr) { return 0; }
    std::lock_guard<std::mutex> lk(g_scratch_mu);
    if (g_scratch_ptr[dev].load() != nullptr) { return 0; }
    double *ptr = nullptr;
    HIP_TRY(fn, hipGetSymbolAddress((void **)&ptr, HIP_SYMBOL(scratch)));
    if (!ptr) { return fail(fn, "hipGetSymbolAddress returned a null address for the static scratch"); }
    g_scratch_ptr[dev].store(ptr);
    return 0;
}

int batch_run(speech_batch *b, const void *X, const short *X16, void *Y, void *stream, float *kernel_ms)
{
    const char *fn = "speech_batch_run";
    if (!b || !b->plan) { return fail(fn, "null batch"); }
    speech_plan *p = b->plan;
    if (b->tiles.empty()) { if (kernel_ms) { *kernel_ms = 0.0f; } return 0; }
    if ((!X && !X16) || !Y) { return fail(fn, "null device pointer"); }
    DeviceGuard g(p->device);
    if (!g.ok) { return fail(fn, "cannot select device %d", p->device); }
    hipStream_t s = (hipStream_t)stream;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (kernel_ms)
    {
        if (!b->ev0) { HIP_TRY(fn, hipEventCreate(&b->ev0)); HIP_TRY(fn, hipEventCreate(&b->ev1)); }
        e0 = b->ev0; e1 = b->ev1;
    }
    if (p->o.dtype == SPEECH_F64)
    {
        FrameParams<double> fp; fill_params<double>(b, X, X16, Y, &fp);
        if (e0) { HIP_TRY(fn, hipEventRecord(e0, s)); }
        if (p->quad64) { HIP_TRY(fn, launch_quad_f64(fp, s)); } else { HIP_TRY(fn, launch_generic_f64(fp, s)); }
        if (e1) { HIP_TRY(fn, hipEventRecord(e1, s)); }
    }
    else
    {
        FrameParams<float> fp; fill_params<float>(b, X, X16, Y, &fp);
        if (p->tuned) { HIP_TRY(fn, launch_fast(p->fast, fp, s, e0, e1)); }
        else
        {
            if (e0) { HIP_TRY(fn, hipEventRecord(e0, s)); }
            HIP_TRY(fn, launch_generic_f32(fp, s));
            if (e1) { HIP_TRY(fn, hipEventRecord(e1, s)); }
        }
    }
    if (p->o.mn0)
    {
        // per-feature mean over the utterance's frames, D wide (intended behaviour of c/kaldi_fbank.c:355-370, DESIGN.md "mn0"):
        // the coalesced moments + affine passes of the post-processing kernels, mean accumulated in double
        PostParams pp; uint32_t mw;
        if (post_begin(fn, b, &pp, &mw)) { return 1; }
        if (post_moments(fn, b, Y, &pp, s)) { return 1; }
        pp.Y = Y; pp.mode = 1; pp.norm_vars = 0;
        HIP_TRY(fn, launch_post(POST_AFFINE, p->o.dtype == SPEECH_F64, pp, mw, s));
    }
    if (kernel_ms)
    {
        HIP_TRY(fn, hipEventSynchronize(e1));
        HIP_TRY(fn, hipStreamSynchronize(s));
        HIP_TRY(fn, hipEventElapsedTime(kernel_ms, e0, e1));
    }
    return 0;
}

}  // namespace

extern "C" {

void speech_opts_default(speech_opts *o, int kind)
{
    memset(o, 0, sizeof(*o));
    o->kind = kind; o->dtype = SPEECH_F32;
    o->sr = 16000.0; o->frame_length = 25.0; o->frame_shift = 10.0;        // srci/kaldi_mfcc.cpp:106-115
    o->snip_edges = 0; o->dither = 0.0; o->dc0 = 0; o->raw_energy = 0;     // srci/kaldi_mfcc.cpp:118-128 (CLI dither default is 0.1)
    o->preemph = 0.97; strcpy(o->win_type, "povey");                       // srci/kaldi_mfcc.cpp:131-135
    o->amp = 0; o->lof = 20.0; o->hif = 8000.0; o->B = 23; o->lg = 1;      // srci/kaldi_mfcc.cpp:144-156 (hif = Nyquist)
    o->use_energy = 0; o->C = 13; o->Q = 22.0;                             // srci/kaldi_mfcc.cpp:159-169
    o->compress = 1.0 / 3.0; o->cep_scale = 1.0; o->lpc_order = 0; o->mn0 = 0;
    o->dither_seed = 0x5EED2026ull; o->force_generic = 0;
}

int speech_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { return 0; }
    return n;
}

const char *speech_last_error(void) { return g_err.c_str(); }

int speech_plan_create(speech_plan **out, const speech_opts *opts, int device)
{
    const char *fn = "speech_plan_create";
    if (!out || !opts) { return fail(fn, "null argument"); }
    *out = nullptr;
    speech_opts o = *opts;
    o.win_type[sizeof(o.win_type) - 1] = '\0';
    Sizes sz;
    if (o.dtype == SPEECH_F64 ? validate<double>(o, &sz) : validate<float>(o, &sz)) { return 1; }
    if (!window_known(o.win_type))
    {
        return fail(kind_name(o.kind, o.dtype), "window type must be rectangular, blackman, hamming, hann or povey (lower-cased)");
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) { return fail(fn, "no HIP device available (this library has no CPU fallback)"); }
    if (device < 0 || device >= ndev) { return fail(fn, "device %d out of range (0..%d)", device, ndev - 1); }
    speech_plan *p = new (std::nothrow) speech_plan();
    if (!p) { return fail(fn, "out of host memory"); }
    p->o = o; p->device = device; p->sz = sz; p->tuned = false; p->quad64 = false; p->fast = nullptr;
    p->P = (o.kind == SPEECH_PLP) ? ((o.lpc_order > 0u) ? o.lpc_order : o.C - 1u) : 0u;     // order = C-1
    p->d_win = p->d_twr = p->d_twi = p->d_f2b = p->d_dct = p->d_lift = p->d_eql = p->d_idft = nullptr;
    p->d_beg = p->d_num = p->d_off = nullptr;
    if (p->P > 256u) { delete p; return fail(fn, "lpc_order above 256 is not supported"); }
    DeviceGuard g(device);
    if (!g.ok) { delete p; return fail(fn, "cannot select device %d", device); }
    if (scratch_init(fn, device)) { delete p; return 1; }                  // once per device: the address of the static scratch
    const int rc = (o.dtype == SPEECH_F64) ? plan_build<double>(p) : plan_build<float>(p);
    if (rc) { speech_plan_destroy(p); return 1; }
    if (!p->tuned && !p->quad64)
    {
        const size_t lds = generic_lds_bytes((int)sz.nfft, (int)o.B, (int)o.C, (int)p->P, o.dtype == SPEECH_F64 ? 8 : 4);
        if (lds > GENERIC_MAX_LDS) { speech_plan_destroy(p); return fail(fn, "frame of %zu samples does not fit in LDS (%zu bytes needed)", sz.L, lds); }
        const hipError_t e = generic_prepare((int)sz.nfft, o.dtype == SPEECH_F64 ? 8u : 4u, lds);
        if (e != hipSuccess) { speech_plan_destroy(p); return fail(fn, "cannot reserve %zu bytes of LDS for the frame kernel: %s", lds, hipGetErrorString(e)); }
    }
    {
        const hipError_t e = post_prepare(o.dtype == SPEECH_F64, (int)sz.D);     // mn0 / cmvn / affine passes of this plan's batches
        if (e != hipSuccess) { speech_plan_destroy(p); return fail(fn, "cannot reserve LDS for the normalisation pass: %s", hipGetErrorString(e)); }
    }
    *out = p;
    return 0;
}

void speech_plan_destroy(speech_plan *p)
{
    if (!p) { return; }
    DeviceGuard g(p->device);
    for (void *d : p->dev_allocs) { (void)hipFree(d); }
    if (p->fast) { fast_tables_destroy(p->fast); }
    delete p;
}

int speech_plan_sizes(const speech_plan *p, size_t *L, size_t *stp, size_t *nfft, size_t *D, int *tuned)
{
    if (!p) { return fail("speech_plan_sizes", "null plan"); }
    if (L) { *L = p->sz.L; } if (stp) { *stp = p->sz.stp; } if (nfft) { *nfft = p->sz.nfft; } if (D) { *D = p->sz.D; }
    if (tuned) { *tuned = p->tuned ? 1 : (p->quad64 ? 2 : 0); }
    return 0;
}

int speech_plan_num_frames(const speech_plan *p, size_t N, size_t *W)
{
    if (W) { *W = 0; }
    if (!p) { return 1; }
    if (N < 1u || N > 2147483647u) { return 1; }                       // int-indexed in the reference: c/kaldi_mfcc.c:82-84
    if (p->o.snip_edges && p->sz.L > N) { return 1; }                  // c/kaldi_mfcc.c:70
    const size_t w = p->o.snip_edges ? 1u + (N - p->sz.L) / p->sz.stp : (N + p->sz.stp / 2u) / p->sz.stp;   // c/kaldi_mfcc.c:77
    if (W) { *W = w; }
    return 0;
}

int speech_opts_frames(const speech_opts *opts, size_t N, size_t *W, size_t *D)
{
    const char *fn = "speech_opts_frames";
    if (W) { *W = 0; }
    if (!opts) { return fail(fn, "null options"); }
    speech_opts o = *opts;
    o.win_type[sizeof(o.win_type) - 1] = '\0';
    Sizes sz;
    if (o.dtype == SPEECH_F64 ? validate<double>(o, &sz) : validate<float>(o, &sz)) { return 1; }
    const char *kfn = kind_name(o.kind, o.dtype);
    if (!window_known(o.win_type)) { return fail(kfn, "window type must be rectangular, blackman, hamming, hann or povey (lower-cased)"); }
    if (D) { *D = sz.D; }
    if (N < 1u) { return fail(kfn, "N (nsamps in signal) must be positive"); }
    if (N > 2147483647u) { return fail(kfn, "%zu samples not supported (int-indexed)", N); }              // c/kaldi_mfcc.c:82-84
    if (o.snip_edges && sz.L > N) { return fail(kfn, "frame length must be < signal length if snip_edges"); }   // c/kaldi_mfcc.c:70
    if (W) { *W = o.snip_edges ? 1u + (N - sz.L) / sz.stp : (N + sz.stp / 2u) / sz.stp; }                 // c/kaldi_mfcc.c:77
    return 0;
}

int speech_batch_create(speech_batch **out, speech_plan *p, size_t nutt, const size_t *n_samples, const size_t *x_off, const size_t *y_off)
{
    return speech::batch_create(out, p, nutt, n_samples, x_off, y_off, 0u, nullptr);
}

}  // extern "C"

// Host-side geometry of a batch: utterance and tile descriptors, totals.  Utterance u draws its dither as utterance number
// uids[u], or uid_base + u when uids is null.  Fills `b` (whose vectors keep their capacity: the drop-in path reuses one per
// thread); touches no device memory.
int speech::batch_fill(speech_batch *b, speech_plan *p, size_t nutt, const size_t *n_samples, const size_t *x_off, const size_t *y_off, uint32_t uid_base,
                       const unsigned int *uids)
{
    const char *fn = "speech_batch_create";
    b->plan = p; b->d_utts = nullptr; b->d_tiles = nullptr; b->d_mom = nullptr; b->d_zrow = nullptr; b->zrow_n = b->zrow_off = -1; b->yoff_pow2 = -1; b->ev0 = b->ev1 = nullptr;
    b->d_ids = nullptr; b->max_n = 0; b->max_w = 0; b->d_plp_en = nullptr; b->plp_rows = 0;
    const bool plp_en = p->tuned && p->o.kind == SPEECH_PLP && p->o.use_energy;      // fast_mel.hip: energies travel between the two launches
    b->total_samples = b->total_out = b->total_frames = 0;
    b->tiles.clear(); b->zrow.clear();
    const int tile = p->tuned ? fast_tile_frames(p->o, p->sz.L, p->sz.nfft) : (p->quad64 ? QUAD_F64_TILE_FRAMES : GENERIC_TILE_FRAMES);
    const char *kfn = kind_name(p->o.kind, p->o.dtype);
    b->utts.resize(nutt);
    size_t xo = 0, yo = 0;
    for (size_t u = 0; u < nutt; ++u)
    {
        const size_t N = n_samples[u];
        size_t W = 0;
        if (N < 1u) { return fail(kfn, "N (nsamps in signal) must be positive"); }
        if (p->o.snip_edges && p->sz.L > N) { return fail(kfn, "frame length must be < signal length if snip_edges"); }
        if (speech_plan_num_frames(p, N, &W)) { return fail(kfn, "utterance %zu: %zu samples not supported (int-indexed)", u, N); }
        UttDesc d;
        d.x_off = x_off ? x_off[u] : xo; d.n = N; d.y_off = y_off ? y_off[u] : yo; d.W = W;
        b->utts[u] = d;
        xo += N; yo += W * p->sz.D;
        b->total_samples += N; b->total_out += W * p->sz.D; b->total_frames += W;
        if (N > b->max_n) { b->max_n = N; }
        if (W > b->max_w) { b->max_w = W; }
        if (plp_en) { b->plp_rows = std::max(b->plp_rows, (size_t)(d.y_off / p->sz.D) + W); }
        for (size_t f0 = 0; f0 < W; f0 += (size_t)tile)
        {
            TileDesc t; t.utt = (uint32_t)u; t.frame0 = (uint32_t)f0; t.nframes = (uint32_t)((W - f0 < (size_t)tile) ? W - f0 : (size_t)tile); t.uid = uids ? (uint32_t)uids[u] : uid_base + (uint32_t)u;
            b->tiles.push_back(t);
        }
    }
    if (b->tiles.size() > 0x7FFFFFFFull) { return fail(fn, "batch too large for one launch (%zu tiles)", b->tiles.size()); }
    return 0;
}

int speech::batch_create(speech_batch **out, speech_plan *p, size_t nutt, const size_t *n_samples, const size_t *x_off, const size_t *y_off, uint32_t uid_base,
                         const unsigned int *uids)
{
    const char *fn = "speech_batch_create";
    if (!out || !p || (nutt && !n_samples)) { return fail(fn, "null argument"); }
    *out = nullptr;
    if (nutt > 0xFFFFFFFFull) { return fail(fn, "too many utterances"); }
    speech_batch *b = new (std::nothrow) speech_batch();
    if (!b) { return fail(fn, "out of host memory"); }
    if (batch_fill(b, p, nutt, n_samples, x_off, y_off, uid_base, uids)) { delete b; return 1; }
    DeviceGuard g(p->device);
    if (!g.ok) { delete b; return fail(fn, "cannot select device %d", p->device); }
    hipError_t e = hipSuccess;
    if (nutt) { e = hipMalloc((void **)&b->d_utts, nutt * sizeof(UttDesc)); }
    if (e == hipSuccess && nutt) { e = hipMemcpy(b->d_utts, b->utts.data(), nutt * sizeof(UttDesc), hipMemcpyHostToDevice); }
    if (e == hipSuccess && !b->tiles.empty()) { e = hipMalloc((void **)&b->d_tiles, b->tiles.size() * sizeof(TileDesc)); }
    if (e == hipSuccess && !b->tiles.empty()) { e = hipMemcpy(b->d_tiles, b->tiles.data(), b->tiles.size() * sizeof(TileDesc), hipMemcpyHostToDevice); }
    // mn0 plans run the moments pass inside speech_batch_run: its buffer belongs to batch creation, so that the run itself
    // only launches (and can be captured in a HIP graph)
    if (e == hipSuccess && nutt && p->o.mn0) { e = hipMalloc((void **)&b->d_mom, mom_doubles(nutt, p->sz.D) * sizeof(double)); }
    // likewise the energy scratch of a tuned plp plan that uses the energy (one float per output row)
    if (e == hipSuccess && b->plp_rows) { e = hipMalloc((void **)&b->d_plp_en, b->plp_rows * sizeof(float)); }
    if (e != hipSuccess) { speech_batch_destroy(b); return fail(fn, "device allocation failed: %s", hipGetErrorString(e)); }
    *out = b;
    return 0;
}

void speech::plan_info(const speech_plan *p, int *device, int *dtype, size_t *D)
{
    if (device) { *device = p->device; }
    if (dtype) { *dtype = p->o.dtype; }
    if (D) { *D = p->sz.D; }
}

extern "C" {

void speech_batch_destroy(speech_batch *b)
{
    if (!b) { return; }
    DeviceGuard g(b->plan ? b->plan->device : 0);
    if (b->d_utts) { (void)hipFree(b->d_utts); }
    if (b->d_tiles) { (void)hipFree(b->d_tiles); }
    if (b->d_mom) { (void)hipFree(b->d_mom); }
    if (b->d_plp_en) { (void)hipFree(b->d_plp_en); }
    if (b->d_zrow) { (void)hipFree(b->d_zrow); }
    if (b->d_ids) { (void)hipFree(b->d_ids); }
    if (b->ev0) { (void)hipEventDestroy(b->ev0); }
    if (b->ev1) { (void)hipEventDestroy(b->ev1); }
    delete b;
}

int speech_batch_totals(const speech_batch *b, size_t *total_samples, size_t *total_out, size_t *total_frames)
{
    if (!b) { return fail("speech_batch_totals", "null batch"); }
    if (total_samples) { *total_samples = b->total_samples; }
    if (total_out) { *total_out = b->total_out; }
    if (total_frames) { *total_frames = b->total_frames; }
    return 0;
}

int speech_batch_offsets(const speech_batch *b, size_t *x_off, size_t *y_off, size_t *n_frames)
{
    if (!b) { return fail("speech_batch_offsets", "null batch"); }
    for (size_t u = 0; u < b->utts.size(); ++u)
    {
        if (x_off) { x_off[u] = (size_t)b->utts[u].x_off; }
        if (y_off) { y_off[u] = (size_t)b->utts[u].y_off; }
        if (n_frames) { n_frames[u] = (size_t)b->utts[u].W; }
    }
    return 0;
}

int speech_batch_set_ids(speech_batch *b, const unsigned int *ids)
{
    const char *fn = "speech_batch_set_ids";
    if (!b || !b->plan) { return fail(fn, "null batch"); }
    for (TileDesc &t : b->tiles) { t.uid = ids ? (uint32_t)ids[t.utt] : t.utt; }
    if (b->tiles.empty()) { return 0; }
    DeviceGuard g(b->plan->device);
    if (!g.ok) { return fail(fn, "cannot select device %d", b->plan->device); }
    HIP_TRY(fn, hipDeviceSynchronize());                                     // a run in flight still reads the old list
    HIP_TRY(fn, hipMemcpy(b->d_tiles, b->tiles.data(), b->tiles.size() * sizeof(TileDesc), hipMemcpyHostToDevice));
    return 0;
}

int speech_batch_run(speech_batch *b, const void *X, void *Y, void *stream) { return batch_run(b, X, nullptr, Y, stream, nullptr); }

int speech_batch_run_pcm16(speech_batch *b, const short *X16, void *Y, void *stream, float *kernel_ms)
{
    if (!X16) { return fail("speech_batch_run_pcm16", "null device pointer"); }
    return batch_run(b, nullptr, X16, Y, stream, kernel_ms);
}

int speech_batch_run_timed(speech_batch *b, const void *X, void *Y, void *stream, float *kernel_ms)
{
    if (!kernel_ms) { return fail("speech_batch_run_timed", "null kernel_ms"); }
    return batch_run(b, X, nullptr, Y, stream, kernel_ms);
}

// ---- corpus post-processing on the packed features; generalises mn0, c/kaldi_fbank.c:355-370 ----

int speech_batch_stats(speech_batch *b, const void *Y, double *sum, double *sumsq, unsigned long long *frames, void *stream)
{
    const char *fn = "speech_batch_stats";
    PostParams pp; uint32_t mw;
    if (post_begin(fn, b, &pp, &mw)) { return 1; }
    if (!sum || !sumsq) { return fail(fn, "null output"); }
    const int D = pp.D;
    for (int d = 0; d < D; ++d) { sum[d] = 0.0; sumsq[d] = 0.0; }
    if (frames) { *frames = (unsigned long long)b->total_frames; }
    if (pp.nutt == 0 || b->total_frames == 0) { return 0; }
    if (!Y) { return fail(fn, "null device pointer"); }
    DeviceGuard g(b->plan->device);
    if (!g.ok) { return fail(fn, "cannot select device %d", b->plan->device); }
    hipStream_t s = (hipStream_t)stream;
    if (post_moments(fn, b, Y, &pp, s)) { return 1; }
    // a tree of fixed shape over the utterances (groups of 64 in order, on the device; the last <= 64 rows on the host):
    // the same sums on every run, and only a few KB cross PCIe
    size_t n = pp.nutt;
    const double *src = b->d_mom;
    double *dst = b->d_mom + ((size_t)pp.nutt + 1u) * 2u * (size_t)D;
    while (n > (size_t)POST_REDUCE_GROUP)
    {
        HIP_TRY(fn, launch_reduce_rows(src, dst, n, 2 * D, s));
        n = (n + POST_REDUCE_GROUP - 1) / POST_REDUCE_GROUP;
        src = dst; dst += n * 2u * (size_t)D;
    }
    std::vector<double> h(n * 2u * (size_t)D);
    HIP_TRY(fn, hipMemcpyAsync(h.data(), src, h.size() * sizeof(double), hipMemcpyDeviceToHost, s));
    HIP_TRY(fn, hipStreamSynchronize(s));
    for (size_t u = 0; u < n; ++u)
    {
        for (int d = 0; d < D; ++d) { sum[d] += h[u * 2 * D + d]; sumsq[d] += h[u * 2 * D + D + d]; }
    }
    return 0;
}

int speech_batch_cmvn(speech_batch *b, void *Y, int norm_vars, void *stream)
{
    const char *fn = "speech_batch_cmvn";
    PostParams pp; uint32_t mw;
    if (post_begin(fn, b, &pp, &mw)) { return 1; }
    if (pp.nutt == 0 || b->total_frames == 0) { return 0; }
    if (!Y) { return fail(fn, "null device pointer"); }
    DeviceGuard g(b->plan->device);
    if (!g.ok) { return fail(fn, "cannot select device %d", b->plan->device); }
    hipStream_t s = (hipStream_t)stream;
    if (post_moments(fn, b, Y, &pp, s)) { return 1; }
    pp.Y = Y; pp.mode = 1; pp.norm_vars = norm_vars ? 1 : 0;
    HIP_TRY(fn, launch_post(POST_AFFINE, b->plan->o.dtype == SPEECH_F64, pp, mw, s));
    return 0;
}

int speech_batch_cmvn_noalloc(speech_batch *b, void *Y, int norm_vars, void *stream)
{
    const char *fn = "speech_batch_cmvn_noalloc";
    PostParams pp; uint32_t mw;
    if (post_begin(fn, b, &pp, &mw)) { return 1; }
    if (pp.nutt == 0 || b->total_frames == 0) { return 0; }
    if (!Y) { return fail(fn, "null device pointer"); }
    const size_t need = (size_t)pp.nutt * 2u * (size_t)pp.D;
    if (need > SCRATCH_DOUBLES)
    {
        return fail(fn, "%u utterances x %d features need %zu bytes of moments; the static scratch holds %zu (use speech_batch_cmvn)",
                    pp.nutt, pp.D, need * sizeof(double), SCRATCH_DOUBLES * sizeof(double));
    }
    const int dev = b->plan->device;
    double *ptr = (dev >= 0 && dev < SCRATCH_MAX_DEVICES) ? g_scratch_ptr[dev].load() : nullptr;
    if (!ptr) { return fail(fn, "no static scratch on device %d", dev); }
    DeviceGuard g(dev);
    if (!g.ok) { return fail(fn, "cannot select device %d", dev); }
    hipStream_t s = (hipStream_t)stream;
    // the moments pass writes every entry the affine pass reads: nothing has to start at zero
    pp.Yin = Y; pp.mom = ptr;
    HIP_TRY(fn, launch_post(POST_MOMENTS, b->plan->o.dtype == SPEECH_F64, pp, 0, s));
    pp.Y = Y; pp.mode = 1; pp.norm_vars = norm_vars ? 1 : 0;
    HIP_TRY(fn, launch_post(POST_AFFINE, b->plan->o.dtype == SPEECH_F64, pp, mw, s));
    return 0;
}

int speech_batch_synth_audio(speech_batch *b, void *X, int pcm16, unsigned long long seed, const unsigned long long *ids, void *stream)
{
    const char *fn = "speech_batch_synth_audio";
    if (!b || !b->plan) { return fail(fn, "null batch"); }
    if (b->utts.empty()) { return 0; }
    if (!X) { return fail(fn, "null device pointer"); }
    DeviceGuard g(b->plan->device);
    if (!g.ok) { return fail(fn, "cannot select device %d", b->plan->device); }
    hipStream_t s = (hipStream_t)stream;
    if (ids)
    {
        if (!b->d_ids) { HIP_TRY(fn, hipMalloc((void **)&b->d_ids, b->utts.size() * sizeof(unsigned long long))); }
        HIP_TRY(fn, hipMemcpyAsync(b->d_ids, ids, b->utts.size() * sizeof(unsigned long long), hipMemcpyHostToDevice, s));
        HIP_TRY(fn, hipStreamSynchronize(s));                                // ids may be pageable and leave scope
    }
    const int elem = pcm16 ? 2 : (b->plan->o.dtype == SPEECH_F64 ? 8 : 4);
    HIP_TRY(fn, launch_synth(X, elem, b->d_utts, ids ? b->d_ids : nullptr, (uint32_t)b->utts.size(), b->max_n, seed, s));
    return 0;
}

int speech_synth_audio_host(unsigned long long seed, unsigned long long utt, size_t n, short *out)
{
    if (!out && n) { return fail("speech_synth_audio_host", "null pointer"); }
    const SynthUtt s = synth_utt(seed, utt);
    for (size_t i = 0; i < n; ++i) { out[i] = (short)synth_sample(s, (unsigned)i); }
    return 0;
}

int speech_batch_affine(speech_batch *b, void *Y, const double *scale, const double *offset, void *stream)
{
    const char *fn = "speech_batch_affine";
    PostParams pp; uint32_t mw;
    if (post_begin(fn, b, &pp, &mw)) { return 1; }
    if (!scale || !offset) { return fail(fn, "null scale / offset"); }
    if (pp.nutt == 0 || b->total_frames == 0) { return 0; }
    if (!Y) { return fail(fn, "null device pointer"); }
    DeviceGuard g(b->plan->device);
    if (!g.ok) { return fail(fn, "cannot select device %d", b->plan->device); }
    hipStream_t s = (hipStream_t)stream;
    const size_t D = (size_t)pp.D;
    if (!b->d_mom) { HIP_TRY(fn, hipMalloc((void **)&b->d_mom, mom_doubles(pp.nutt, D) * sizeof(double))); }
    double *d_aff = b->d_mom + (size_t)pp.nutt * 2 * D;
    std::vector<double> h(2 * D);
    for (size_t d = 0; d < D; ++d) { h[d] = scale[d]; h[D + d] = offset[d]; }
    HIP_TRY(fn, hipMemcpyAsync(d_aff, h.data(), h.size() * sizeof(double), hipMemcpyHostToDevice, s));
    HIP_TRY(fn, hipStreamSynchronize(s));                                    // h leaves scope
    pp.Y = Y; pp.mode = 0; pp.scale = d_aff; pp.offset = d_aff + D;
    HIP_TRY(fn, launch_post(POST_AFFINE, b->plan->o.dtype == SPEECH_F64, pp, mw, s));
    return 0;
}

int speech_batch_add_deltas(speech_batch *b, const void *Y, void *Z, int order, int window, void *stream)
{
    const char *fn = "speech_batch_add_deltas";
    PostParams pp; uint32_t mw;
    if (post_begin(fn, b, &pp, &mw)) { return 1; }
    if (order < 0 || order > POST_MAX_ORDER) { return fail(fn, "order must be in [0 %d]", POST_MAX_ORDER); }
    if (window < 1 || window > POST_MAX_WINDOW) { return fail(fn, "window must be in [1 %d]", POST_MAX_WINDOW); }
    if (pp.nutt == 0 || b->total_frames == 0) { return 0; }
    if (!Y || !Z) { return fail(fn, "null device pointer"); }
    if (Y == Z) { return fail(fn, "add_deltas cannot run in place"); }
    DeviceGuard g(b->plan->device);
    if (!g.ok) { return fail(fn, "cannot select device %d", b->plan->device); }
    // filters as Kaldi's DeltaFeatures: scales[0] = {1}; scales[k] = scales[k-1] convolved with j / sum(j^2), j = -window..window
    pp.dscale[0][0] = 1.0;
    for (int k = 1; k <= order; ++k)
    {
        const int ph = (k - 1) * window, ch = k * window;
        double norm = 0.0;
        for (int j = -window; j <= window; ++j)
        {
            norm += (double)(j * j);
            for (int i = -ph; i <= ph; ++i) { pp.dscale[k][j + i + ch] += (double)j * pp.dscale[k - 1][i + ph]; }
        }
        for (int i = 0; i <= 2 * ch; ++i) { pp.dscale[k][i] /= norm; }
    }
    pp.Yin = Y; pp.Y = Z; pp.order = order; pp.window = window;
    HIP_TRY(fn, launch_post(POST_DELTAS, b->plan->o.dtype == SPEECH_F64, pp, mw, (hipStream_t)stream));
    return 0;
}

int speech_batch_splice(speech_batch *b, const void *Y, void *Z, int left, int right, void *stream)
{
    const char *fn = "speech_batch_splice";
    PostParams pp; uint32_t mw;
    if (post_begin(fn, b, &pp, &mw)) { return 1; }
    if (left < 0 || right < 0 || left > 64 || right > 64) { return fail(fn, "left and right context must be in [0 64]"); }
    if (pp.nutt == 0 || b->total_frames == 0) { return 0; }
    if (!Y || !Z) { return fail(fn, "null device pointer"); }
    if (Y == Z) { return fail(fn, "splice cannot run in place"); }
    DeviceGuard g(b->plan->device);
    if (!g.ok) { return fail(fn, "cannot select device %d", b->plan->device); }
    pp.Yin = Y; pp.Y = Z; pp.left = left; pp.right = right; pp.vec16 = splice_vec16(b, Y, Z);
    HIP_TRY(fn, launch_post(POST_SPLICE, b->plan->o.dtype == SPEECH_F64, pp, mw, (hipStream_t)stream));
    return 0;
}

int speech_batch_subsample_rows(const speech_batch *b, int n, int offset, size_t *rows, size_t *total_rows)
{
    const char *fn = "speech_batch_subsample_rows";
    if (!b || !b->plan) { return fail(fn, "null batch"); }
    if (n < 1 || offset < 0 || offset >= n) { return fail(fn, "need n >= 1 and 0 <= offset < n"); }
    size_t tot = 0;
    for (size_t u = 0; u < b->utts.size(); ++u)
    {
        const size_t W = (size_t)b->utts[u].W, r = W > (size_t)offset ? (W - (size_t)offset + (size_t)n - 1u) / (size_t)n : 0u;
        if (rows) { rows[u] = r; }
        tot += r;
    }
    if (total_rows) { *total_rows = tot; }
    return 0;
}

int speech_batch_splice_subsample(speech_batch *b, const void *Y, void *Z, int left, int right, int n, int offset, void *stream)
{
    const char *fn = "speech_batch_splice_subsample";
    PostParams pp; uint32_t mw;
    if (post_begin(fn, b, &pp, &mw)) { return 1; }
    if (left < 0 || right < 0 || left > 64 || right > 64) { return fail(fn, "left and right context must be in [0 64]"); }
    if (n < 1 || offset < 0 || offset >= n) { return fail(fn, "need n >= 1 and 0 <= offset < n"); }
    if (pp.nutt == 0 || b->total_frames == 0) { return 0; }
    if (!Y || !Z) { return fail(fn, "null device pointer"); }
    if (Y == Z) { return fail(fn, "splice cannot run in place"); }
    DeviceGuard g(b->plan->device);
    if (!g.ok) { return fail(fn, "cannot select device %d", b->plan->device); }
    hipStream_t s = (hipStream_t)stream;
    if (!b->d_zrow) { HIP_TRY(fn, hipMalloc((void **)&b->d_zrow, b->utts.size() * sizeof(unsigned long long))); }
    if (b->zrow_n != n || b->zrow_off != offset)
    {
        HIP_TRY(fn, hipStreamSynchronize(s));                                // an earlier call on this stream may still read the table
        b->zrow.resize(b->utts.size());
        unsigned long long acc = 0;
        for (size_t u = 0; u < b->utts.size(); ++u)
        {
            b->zrow[u] = acc;
            const unsigned long long W = b->utts[u].W;
            acc += W > (unsigned long long)offset ? (W - (unsigned long long)offset + (unsigned long long)n - 1u) / (unsigned long long)n : 0u;
        }
        HIP_TRY(fn, hipMemcpy(b->d_zrow, b->zrow.data(), b->zrow.size() * sizeof(unsigned long long), hipMemcpyHostToDevice));
        b->zrow_n = n; b->zrow_off = offset;
    }
    pp.Yin = Y; pp.Y = Z; pp.left = left; pp.right = right; pp.sub_n = n; pp.sub_off = offset; pp.zrow = b->d_zrow; pp.vec16 = splice_vec16(b, Y, Z);
    HIP_TRY(fn, launch_post(POST_SPLICE_SUB, b->plan->o.dtype == SPEECH_F64, pp, (mw + (uint32_t)n - 1u) / (uint32_t)n, s));
    return 0;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------
// The ten drop-in entry points (include/codee_speech.h).  Host pointers in, host pointers out.
// ---------------------------------------------------------------------------------------------
namespace {

// The reference rebuilds its tables on every call (c/kaldi_mfcc.c:105-210) and frees all it allocated before it returns
// (:355-357).  Here a call looks its option set up in a small process-wide cache of plans (plans are immutable once built,
// so threads share them) and stages through per-thread buffers that only grow, so that repeated calls cost two copies and
// one launch.  Cache entries and calls in flight hold the plan by shared_ptr: an evicted plan is destroyed, with its device
// tables, as soon as the last call that uses it has returned.
struct PlanCacheEntry { speech_opts o; int device; std::shared_ptr<speech_plan> plan; unsigned long long stamp; };
// dither of the drop-in calls (speech_set_dropin_dither): seed, pinned utterance number (< 0: none), call counter
std::atomic<unsigned long long> g_dropin_seed{0x5EED2026ull};
std::atomic<long long> g_dropin_utt{-1};
std::atomic<uint32_t> g_dropin_counter{0u};
std::mutex g_cache_mu;
std::vector<PlanCacheEntry> g_cache;
unsigned long long g_cache_clock = 0;
constexpr size_t PLAN_CACHE_MAX = 16;

bool same_opts(const speech_opts &a, const speech_opts &b)
{
    return a.kind == b.kind && a.dtype == b.dtype && a.sr == b.sr && a.frame_length == b.frame_length && a.frame_shift == b.frame_shift &&
           a.snip_edges == b.snip_edges && a.dither == b.dither && a.dc0 == b.dc0 && a.raw_energy == b.raw_energy && a.preemph == b.preemph &&
           strncmp(a.win_type, b.win_type, sizeof(a.win_type)) == 0 && a.amp == b.amp && a.lof == b.lof && a.hif == b.hif && a.B == b.B &&
           a.lg == b.lg && a.use_energy == b.use_energy && a.C == b.C && a.Q == b.Q && a.compress == b.compress && a.cep_scale == b.cep_scale &&
           a.lpc_order == b.lpc_order && a.mn0 == b.mn0 && a.dither_seed == b.dither_seed && a.force_generic == b.force_generic;
}

std::shared_ptr<speech_plan> cached_plan(const speech_opts &o, int dev)
{
    std::shared_ptr<speech_plan> evicted;                                    // destroyed (if this was its last holder) after the lock is gone
    std::lock_guard<std::mutex> lk(g_cache_mu);
    for (auto &e : g_cache) { if (e.device == dev && same_opts(e.o, o)) { e.stamp = ++g_cache_clock; return e.plan; } }
    speech_plan *p = nullptr;
    if (speech_plan_create(&p, &o, dev)) { return nullptr; }
    std::shared_ptr<speech_plan> sp(p, [](speech_plan *q) { speech_plan_destroy(q); });
    if (g_cache.size() >= PLAN_CACHE_MAX)
    {
        size_t lru = 0;
        for (size_t i = 1; i < g_cache.size(); ++i) { if (g_cache[i].stamp < g_cache[lru].stamp) { lru = i; } }
        evicted = std::move(g_cache[lru].plan);
        g_cache.erase(g_cache.begin() + (long)lru);
    }
    g_cache.push_back(PlanCacheEntry{o, dev, sp, ++g_cache_clock});
    return sp;
}

// Per-thread staging of the drop-in calls: device buffers for [descriptors | audio] and the features, a pinned pair for
// small transfers ("stage through a pinned pool"), the mn0 moments, a stream.  Grow-only; freed when the
// thread exits, or when the thread moves to another device.
constexpr size_t PIN_MAX = (size_t)8 << 20;                                  // larger transfers go straight from / to the caller's memory
struct StageBuffers {
    void *dX = nullptr, *dY = nullptr, *hX = nullptr, *hY = nullptr; double *dMom = nullptr; float *dEn = nullptr;
    size_t capX = 0, capY = 0, caphX = 0, caphY = 0, capMom = 0, capEn = 0;
    hipStream_t st = nullptr;
    int device = -1;
    speech_batch batch;                                                      // geometry of the call in flight (vectors keep their capacity)
    void release()
    {
        if (device < 0) { return; }
        DeviceGuard g(device);
        if (st) { (void)hipStreamSynchronize(st); (void)hipStreamDestroy(st); }
        if (dX) { (void)hipFree(dX); } if (dY) { (void)hipFree(dY); } if (dMom) { (void)hipFree(dMom); } if (dEn) { (void)hipFree(dEn); }
        if (hX) { (void)hipHostFree(hX); } if (hY) { (void)hipHostFree(hY); }
        dX = dY = hX = hY = nullptr; dMom = nullptr; dEn = nullptr; st = nullptr;
        capX = capY = caphX = caphY = capMom = capEn = 0; device = -1;
    }
    ~StageBuffers() { release(); }
};

int stage_grow(const char *fn, void **p, size_t *cap, size_t bytes, bool pinned)
{
    if (bytes <= *cap) { return 0; }
    if (*p) { (void)(pinned ? hipHostFree(*p) : hipFree(*p)); *p = nullptr; *cap = 0; }
    const size_t want = bytes + bytes / 4;
    if ((pinned ? hipHostMalloc(p, want, hipHostMallocDefault) : hipMalloc(p, want)) != hipSuccess) { *p = nullptr; return fail(fn, pinned ? "problem with hipHostMalloc" : "problem with hipMalloc"); }
    *cap = want;
    return 0;
}

inline size_t up256(size_t n) { return (n + 255u) & ~(size_t)255; }

// X16 non-null: the audio is 16-bit PCM on the host (speech_extract_pcm16), X unused
template <typename T>
int legacy_call(speech_opts &o, T *Y, const T *X, size_t N, const short *X16 = nullptr)
{
    const char *fn = X16 ? "speech_extract_pcm16" : kind_name(o.kind, o.dtype);
    if (N < 1u) { return fail(fn, "N (nsamps in signal) must be positive"); }                 // c/kaldi_mfcc.c:54
    Sizes sz;
    if (validate<T>(o, &sz)) { return 1; }
    if (o.snip_edges && sz.L > N) { return fail(fn, "frame length must be < signal length if snip_edges"); }   // c/kaldi_mfcc.c:70
    if (N > 2147483647u) { return fail(fn, "N above 2^31-1 is not supported (int-indexed, c/kaldi_mfcc.c:82-84)"); }
    const size_t W = o.snip_edges ? 1u + (N - sz.L) / sz.stp : (N + sz.stp / 2u) / sz.stp;     // c/kaldi_mfcc.c:77
    if (W == 0u) { return 0; }                                                                 // c/kaldi_mfcc.c:78 (before the window name is looked at)
    if (!window_known(o.win_type)) { return fail(fn, "window type must be rectangular, blackman, hamming, hann or povey (lower-cased)"); }
    if ((!X && !X16) || !Y) { return fail(fn, "null pointer"); }
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { return fail(fn, "no HIP device available (this library has no CPU fallback)"); }
    const bool dithered = (T)o.dither > (T)FLT_EPSILON;                      // c/kaldi_mfcc.c:243
    const std::shared_ptr<speech_plan> plan = cached_plan(o, dev);           // held until this call returns
    if (!plan) { return 1; }
    speech_plan *p = plan.get();
    static thread_local StageBuffers sb;
    if (sb.device != dev) { sb.release(); sb.device = dev; }
    if (!sb.st && hipStreamCreateWithFlags(&sb.st, hipStreamNonBlocking) != hipSuccess) { sb.st = nullptr; return fail(fn, "problem creating a stream"); }
    // the reference draws fresh noise on every call (wall-clock seed, c/kaldi_fbank.c:97-101): a dithered drop-in call takes
    // the next number of a process-wide counter as its utterance number, unless the caller pinned one (speech_set_dropin_dither)
    uint32_t uid = 0u;
    if (dithered) { const long long pin = g_dropin_utt.load(); uid = pin >= 0 ? (uint32_t)pin : g_dropin_counter.fetch_add(1u); }
    speech_batch &b = sb.batch;
    if (batch_fill(&b, p, 1u, &N, nullptr, nullptr, uid, nullptr)) { return 1; }
    const size_t nout = W * sz.D, ybytes = nout * sizeof(T);
    const size_t xbytes = X16 ? N * sizeof(short) : N * sizeof(T);
    const size_t tbytes = b.tiles.size() * sizeof(TileDesc);
    const size_t dbytes = up256(sizeof(UttDesc) + tbytes);                   // [UttDesc | tiles] in front of the audio
    const bool pinX = xbytes <= PIN_MAX, pinY = ybytes <= PIN_MAX;
    if (stage_grow(fn, &sb.dX, &sb.capX, dbytes + xbytes, false) || stage_grow(fn, &sb.dY, &sb.capY, ybytes, false) ||
        stage_grow(fn, &sb.hX, &sb.caphX, dbytes + (pinX ? xbytes : 0u), true) || (pinY && stage_grow(fn, &sb.hY, &sb.caphY, ybytes, true))) { return 1; }
    if (p->o.mn0 && stage_grow(fn, (void **)&sb.dMom, &sb.capMom, mom_doubles(1u, sz.D) * sizeof(double), false)) { return 1; }
    if (b.plp_rows && stage_grow(fn, (void **)&sb.dEn, &sb.capEn, b.plp_rows * sizeof(float), false)) { return 1; }
    memcpy(sb.hX, b.utts.data(), sizeof(UttDesc));
    memcpy((char *)sb.hX + sizeof(UttDesc), b.tiles.data(), tbytes);
    const void *xsrc = X16 ? (const void *)X16 : (const void *)X;
    hipError_t e;
    if (pinX)
    {
        memcpy((char *)sb.hX + dbytes, xsrc, xbytes);
        e = hipMemcpyAsync(sb.dX, sb.hX, dbytes + xbytes, hipMemcpyHostToDevice, sb.st);
    }
    else
    {
        e = hipMemcpyAsync(sb.dX, sb.hX, dbytes, hipMemcpyHostToDevice, sb.st);
        if (e == hipSuccess) { e = hipMemcpyAsync((char *)sb.dX + dbytes, xsrc, xbytes, hipMemcpyHostToDevice, sb.st); }
    }
    if (e != hipSuccess) { return fail(fn, "problem copying X to the device (%s)", hipGetErrorString(e)); }
    b.d_utts = (UttDesc *)sb.dX; b.d_tiles = (TileDesc *)((char *)sb.dX + sizeof(UttDesc)); b.d_mom = sb.dMom; b.d_plp_en = b.plp_rows ? sb.dEn : nullptr;     // borrowed: never freed through the batch
    const void *dXa = (const char *)sb.dX + dbytes;
    if (X16 ? speech_batch_run_pcm16(&b, (const short *)dXa, sb.dY, (void *)sb.st, nullptr) : speech_batch_run(&b, dXa, sb.dY, (void *)sb.st)) { return 1; }
    e = hipMemcpyAsync(pinY ? sb.hY : (void *)Y, sb.dY, ybytes, hipMemcpyDeviceToHost, sb.st);
    if (e == hipSuccess) { e = hipStreamSynchronize(sb.st); }
    if (e != hipSuccess) { return fail(fn, "kernel execution or copy of Y failed (%s)", hipGetErrorString(e)); }
    if (pinY) { memcpy(Y, sb.hY, ybytes); }
    return 0;
}

template <typename T>
void front(speech_opts &o, int kind, T sr, T fl, T fs, int snip, T dither, int dc0, int raw_energy, T preemph, const char *win, int mn0)
{
    speech_opts_default(&o, kind);
    o.dither_seed = g_dropin_seed.load();                                    // speech_set_dropin_dither
    o.dtype = std::is_same<T, double>::value ? SPEECH_F64 : SPEECH_F32;
    o.sr = (double)sr; o.frame_length = (double)fl; o.frame_shift = (double)fs; o.snip_edges = snip; o.dither = (double)dither;
    o.dc0 = dc0; o.raw_energy = raw_energy; o.preemph = (double)preemph; o.mn0 = mn0;
    strncpy(o.win_type, win ? win : "", sizeof(o.win_type) - 1); o.win_type[sizeof(o.win_type) - 1] = '\0';
}

}  // namespace

extern "C" void speech_set_dropin_dither(unsigned long long seed, long long utt)
{
    g_dropin_seed.store(seed);
    g_dropin_utt.store(utt);
    if (utt < 0) { g_dropin_counter.store(0u); }
}

extern "C" int speech_extract_pcm16(const speech_opts *opts, const short *X16, size_t N, void *Y)
{
    if (!opts) { return fail("speech_extract_pcm16", "null opts"); }
    if (!X16 && N > 0u) { return fail("speech_extract_pcm16", "null pointer"); }
    speech_opts o = *opts;
    o.win_type[sizeof(o.win_type) - 1] = '\0';
    if (o.dtype == SPEECH_F64) { return legacy_call<double>(o, (double *)Y, nullptr, N, X16); }
    return legacy_call<float>(o, (float *)Y, nullptr, N, X16);
}

namespace codee {
extern "C" {

int kaldi_spectrogram_s(float *Y, float *X, const size_t N, const float sr, const float frame_length, const float frame_shift, const int snip_edges, const float dither, const int dc0, const int raw_energy, const float preemph, const char win_type[], const int mn0)
{
    speech_opts o; front<float>(o, SPEECH_SPECTROGRAM, sr, frame_length, frame_shift, snip_edges, dither, dc0, raw_energy, preemph, win_type, mn0);
    return legacy_call<float>(o, Y, X, N);
}

int kaldi_spectrogram_d(double *Y, double *X, const size_t N, const double sr, const double frame_length, const double frame_shift, const int snip_edges, const double dither, const int dc0, const int raw_energy, const double preemph, const char win_type[], const int mn0)
{
    speech_opts o; front<double>(o, SPEECH_SPECTROGRAM, sr, frame_length, frame_shift, snip_edges, dither, dc0, raw_energy, preemph, win_type, mn0);
    return legacy_call<double>(o, Y, X, N);
}

int kaldi_fbank_s(float *Y, float *X, const size_t N, const float sr, const float frame_length, const float frame_shift, const int snip_edges, const float dither, const int dc0, const int raw_energy, const float preemph, const char win_type[], const int amp, const float lof, const float hif, const size_t B, const int lg, const int use_energy, const int mn0)
{
    speech_opts o; front<float>(o, SPEECH_FBANK, sr, frame_length, frame_shift, snip_edges, dither, dc0, raw_energy, preemph, win_type, mn0);
    o.amp = amp; o.lof = (double)lof; o.hif = (double)hif; o.B = B; o.lg = lg; o.use_energy = use_energy;
    return legacy_call<float>(o, Y, X, N);
}

int kaldi_fbank_d(double *Y, double *X, const size_t N, const double sr, const double frame_length, const double frame_shift, const int snip_edges, const double dither, const int dc0, const int raw_energy, const double preemph, const char win_type[], const int amp, const double lof, const double hif, const size_t B, const int lg, const int use_energy, const int mn0)
{
    speech_opts o; front<double>(o, SPEECH_FBANK, sr, frame_length, frame_shift, snip_edges, dither, dc0, raw_energy, preemph, win_type, mn0);
    o.amp = amp; o.lof = lof; o.hif = hif; o.B = B; o.lg = lg; o.use_energy = use_energy;
    return legacy_call<double>(o, Y, X, N);
}

// The reference's *_default_* functions hard-wire sr=16000, frame_length=25 ms, B=80 and, as written, do not
// work (c/kaldi_fbank.c:36-37; frames stored reversed at :786,804).  Exported for ABI completeness as the
// general path with those constants; parity unpinned.
int kaldi_fbank_default_s(float *Y, float *X, const size_t N, const float frame_shift, const int snip_edges, const float dither, const int dc0, const int raw_energy, const float preemph, const char win_type[], const int amp, const float lof, const float hif, const int lg, const int use_energy, const int mn0)
{
    return kaldi_fbank_s(Y, X, N, 16000.0f, 25.0f, frame_shift, snip_edges, dither, dc0, raw_energy, preemph, win_type, amp, lof, hif, 80u, lg, use_energy, mn0);
}

int kaldi_fbank_default_d(double *Y, double *X, const size_t N, const double frame_shift, const int snip_edges, const double dither, const int dc0, const int raw_energy, const double preemph, const char win_type[], const int amp, const double lof, const double hif, const int lg, const int use_energy, const int mn0)
{
    return kaldi_fbank_d(Y, X, N, 16000.0, 25.0, frame_shift, snip_edges, dither, dc0, raw_energy, preemph, win_type, amp, lof, hif, 80u, lg, use_energy, mn0);
}

int kaldi_mfcc_s(float *Y, float *X, const size_t N, const float sr, const float frame_length, const float frame_shift, const int snip_edges, const float dither, const int dc0, const int raw_energy, const float preemph, const char win_type[], const float lof, const float hif, const size_t B, const size_t C, const float Q, const int use_energy, const int mn0)
{
    speech_opts o; front<float>(o, SPEECH_MFCC, sr, frame_length, frame_shift, snip_edges, dither, dc0, raw_energy, preemph, win_type, mn0);
    o.lof = (double)lof; o.hif = (double)hif; o.B = B; o.C = C; o.Q = (double)Q; o.use_energy = use_energy;
    return legacy_call<float>(o, Y, X, N);
}

int kaldi_mfcc_d(double *Y, double *X, const size_t N, const double sr, const double frame_length, const double frame_shift, const int snip_edges, const double dither, const int dc0, const int raw_energy, const double preemph, const char win_type[], const double lof, const double hif, const size_t B, const size_t C, const double Q, const int use_energy, const int mn0)
{
    speech_opts o; front<double>(o, SPEECH_MFCC, sr, frame_length, frame_shift, snip_edges, dither, dc0, raw_energy, preemph, win_type, mn0);
    o.lof = lof; o.hif = hif; o.B = B; o.C = C; o.Q = Q; o.use_energy = use_energy;
    return legacy_call<double>(o, Y, X, N);
}

int kaldi_plp_s(float *Y, float *X, const size_t N, const float sr, const float frame_length, const float frame_shift, const int snip_edges, const float dither, const int dc0, const int raw_energy, const float preemph, const char win_type[], const float lof, const float hif, const size_t B, const float compress, const size_t C, const float Q, const float cep_scale, const int use_energy, const int mn0)
{
    speech_opts o; front<float>(o, SPEECH_PLP, sr, frame_length, frame_shift, snip_edges, dither, dc0, raw_energy, preemph, win_type, mn0);
    o.lof = (double)lof; o.hif = (double)hif; o.B = B; o.C = C; o.Q = (double)Q; o.use_energy = use_energy;
    o.compress = (double)compress; o.cep_scale = (double)cep_scale; o.lpc_order = 0;
    return legacy_call<float>(o, Y, X, N);
}

int kaldi_plp_d(double *Y, double *X, const size_t N, const double sr, const double frame_length, const double frame_shift, const int snip_edges, const double dither, const int dc0, const int raw_energy, const double preemph, const char win_type[], const double lof, const double hif, const size_t B, const double compress, const size_t C, const double Q, const double cep_scale, const int use_energy, const int mn0)
{
    speech_opts o; front<double>(o, SPEECH_PLP, sr, frame_length, frame_shift, snip_edges, dither, dc0, raw_energy, preemph, win_type, mn0);
    o.lof = lof; o.hif = hif; o.B = B; o.C = C; o.Q = Q; o.use_energy = use_energy;
    o.compress = compress; o.cep_scale = cep_scale; o.lpc_order = 0;
    return legacy_call<double>(o, Y, X, N);
}

}  // extern "C"
}  // namespace codee
