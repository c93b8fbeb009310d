// fast_kernels.hip -- host side of the tuned gfx950 fp32 kernels (fast_spectrogram.hip, fast_mel.hip; their shared
// device code is fast_common.hpp): launch geometry, plan tables, launch.  The kernels serve 512-point frames (257..512
// samples per frame, even frame shift), 256-point frames (129..256 samples), two frames per complex transform, and
// 1024-point frames (513..1024 samples), one frame per 32 lanes.  Everything else runs on generic_kernels.hip.
#include "fast_common.hpp"
#include <mutex>
#include <set>
#include <utility>

namespace speech {

// Row pitch of the mel tile: twice an odd number >= cols, which makes the MFMA A-operand reads conflict-free (a k-step's
// 32 lanes read rows 0..15 at two consecutive columns: banks 2 h r + c with h odd are distinct for r < 16, c < 2)
static int mel_pitch(int cols) { const int h = (cols + 1) / 2; return 2 * (h | 1); }

// LDS a workgroup may use if three are to fit on a CU (160 KB in granules of 1280 bytes)
constexpr size_t LDS_3_PER_CU = 42u * 1280u;

// Launch geometry of a plan on the tuned kernels: 512-point frames (257..512 samples), or 256-point frames (129..256) two
// per transform (`pair`: 8 frames per wave).
static FastGeom fast_geom(const speech_opts &o, size_t L, size_t nfft)
{
    FastGeom g; memset(&g, 0, sizeof(g));
    if (o.dtype != SPEECH_F32 || (nfft != (size_t)NFFT && nfft != (size_t)NFFT / 2u && nfft != 2u * (size_t)NFFT && nfft != 4u * (size_t)NFFT)) { return g; }
    const size_t stp = (size_t)((float)o.sr * (float)o.frame_shift / 1000.0f);
    if (stp < 2u || stp > 2048u || L > nfft || L <= nfft / 2u) { return g; }
    g.pair = (nfft == (size_t)NFFT / 2u);
    g.wide = (nfft == 2u * (size_t)NFFT);
    g.xwide = (nfft == 4u * (size_t)NFFT);
    if (stp > 1024u && !g.xwide) { return g; }
    if ((stp & 1u) && !g.xwide && !g.wide) { return g; }                    // 1024- and 2048-point frames are read sample by sample: any shift; the others read aligned sample pairs
    g.fw = g.xwide ? 1 : (g.wide ? 2 : (g.pair ? 2 * QF : QF));
    g.nz = g.xwide ? (int)((L + 127u) / 128u) : (g.wide ? (int)((L + 63u) / 64u) : (g.pair ? (int)((L + 15u) / 16u) : (int)((L + 31u) / 32u)));
    const size_t F = nfft / 2u + 1u, rows = (g.xwide ? 128u : (g.wide ? 64u : (g.pair ? 16u : 32u))) * (size_t)g.nz;
    const size_t tabf = g.xwide ? (o.kind == SPEECH_SPECTROGRAM ? (size_t)TABX_FLOATS : (size_t)TABX_TWF) : (g.wide ? (size_t)TABW_FLOATS : (size_t)TAB_FLOATS);
    // two workgroups per CU are what the 2048-point kernels are built for (128 VGPRs); the others stay below 64 KB
    const size_t lds_max = g.xwide ? (size_t)100 * 1024u : (size_t)64 * 1024u;   // above 80 KB (long frames with many mel bins): one workgroup per CU
    g.span = (((size_t)g.fw - 1u) * stp + rows + 63u) & ~(size_t)63;        // whole 64-lane load instructions fit the slice
    if (g.xwide) { g.span += 2u * (g.span / 32u); }                        // 2 floats of padding per 32 samples (padx, fast_common.hpp)
    if (o.kind == SPEECH_SPECTROGRAM)
    {
        g.ws = (std::max(g.span + (size_t)g.fw * F, (size_t)TR_WAVE_C) + 3u) & ~(size_t)3;      // + the wave's output, staged behind the parked audio; slices stay 16-byte aligned
        g.tile = (NT / 64) * g.fw * 4;                                      // four sub-tiles
        g.lds = (tabf + (NT / 64) * g.ws) * sizeof(float);
        g.ok = g.lds <= lds_max;
        return g;
    }
    if (o.B < 2u || o.B > 16u * MAX_MTILES) { return g; }                   // 128 mel bins
    size_t D = o.B + (o.use_energy ? 1u : 0u), cols = 0;
    if (o.kind == SPEECH_MFCC) { if (o.C > 96u) { return g; } D = o.C; cols = (o.B + 3u) & ~(size_t)3; }
    if (o.kind == SPEECH_PLP)
    {
        const size_t P = (o.lpc_order > 0u) ? o.lpc_order : o.C - 1u;
        if (P != 12u) { return g; }                                         // the in-register Levinson-Durbin is built for order 12
        if (o.C < P + 1u) { return g; }                                     // the frame's output row carries r[0..12] between the two launches
        D = o.C; cols = (o.B + 2u + 3u) & ~(size_t)3;
    }
    const size_t T = (size_t)MW * (size_t)g.fw;                             // frames per sub-tile: one pass of the workgroup
    const size_t nsub = 1u;
    g.tile = (int)(T * nsub);
    g.ws = std::max(g.span, g.xwide ? (T * PPX + MW - 1) / MW : (g.wide ? (T * PPW + MW - 1) / MW : (g.pair ? (T * PP2 + MW - 1) / MW : (size_t)TR_WAVE)));     // the slices together hold the power tile
    // behind the slices: frame energies | fbank: output tile | mfcc, plp: mel tile | plp: autocorrelations; the mfcc / plp
    // output tile reuses the power tile's place
    const size_t extra = T * nsub + (o.kind == SPEECH_FBANK ? ((T * D + 3u) & ~(size_t)3) : T * (size_t)mel_pitch((int)cols));
    if (T * nsub * D > (size_t)MW * g.ws) { return g; }
    g.ws = (g.ws + 3u) & ~(size_t)3;                                         // slices stay 16-byte aligned
    g.lds = (tabf + MW * g.ws + extra) * sizeof(float);
    g.ok = g.lds <= lds_max;
    // the second contraction's matrix (mfcc: lifter x DCT, K2 x Cp; plp: IDFT bases, K2 x 16) rides in LDS behind the
    // tables when that does not cost a resident workgroup; otherwise the kernel reads it from global memory (L2)
    const size_t m2 = (o.kind == SPEECH_MFCC) ? cols * 16u * ((o.C + 15u) / 16u) : (o.kind == SPEECH_PLP ? cols * 16u : 0u);
    const size_t m2_max = (!g.xwide || g.lds <= (size_t)LDS_3_PER_CU) ? (size_t)LDS_3_PER_CU : (g.lds <= 80u * 1024u ? (size_t)80 * 1024u : lds_max);      // 2048-point plans: three per CU where the frame is short enough, else two up to 80 KB
    if (g.ok && m2 > 0u && g.lds + m2 * sizeof(float) <= m2_max) { g.m2 = m2; g.lds += m2 * sizeof(float); }
    return g;
}

bool fast_supported(const speech_opts &o, size_t L, size_t nfft) { return fast_geom(o, L, nfft).ok; }

int fast_tile_frames(const speech_opts &o, size_t L, size_t nfft) { return fast_geom(o, L, nfft).tile; }

// Opt a kernel in to `bytes` of dynamic LDS, once per (device, kernel).  The attribute belongs to the kernel on a device, not to
// a plan, so every caller asks for the LARGEST size its host layer admits for that kernel (64 KB / 100 KB for the tuned kernels,
// 160 KB for the generic frame kernels, 2 x 4097 doubles for the affine pass): whichever plan is created first, every later
// plan's launch fits.  Done when a plan is created, so that speech_batch_run itself only launches (and can be captured in a
// HIP graph).
hipError_t opt_in_lds_once(const void *k, int bytes)
{
    static std::mutex mu;
    static std::set<std::pair<int, const void *>> opted;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) { return e; }
    std::lock_guard<std::mutex> lock(mu);
    const auto key = std::make_pair(dev, k);
    if (opted.insert(key).second)
    {
        e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        if (e != hipSuccess) { opted.erase(key); }
    }
    return e;
}

static hipError_t opt_in_lds(kernel_fn k, int bytes) { return opt_in_lds_once(reinterpret_cast<const void *>(k), bytes); }

static kernel_fn plan_kernel(const FastTables *t, bool dither, bool pcm)
{
    if (t->kind == SPEECH_SPECTROGRAM) { return fast_spectrogram_kernel_for(t->nz, dither, pcm, t->geom.pair, t->geom.wide, t->geom.xwide); }
    return fast_mel_kernel_for(t->nz, t->kind, dither, t->geom.pair, t->geom.wide, t->geom.xwide);
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
    const FastGeom g = fast_geom(o, L, nfft);
    if (!g.ok) { return nullptr; }
    std::vector<float> tab(g.xwide ? TABX_FLOATS : (g.wide ? TABW_FLOATS : TAB_FLOATS), 0.0f);
    const int o_win = g.xwide ? TABX_WIN : (g.wide ? TABW_WIN : TAB_WIN), o_tw1 = g.xwide ? TABX_TW1 : (g.wide ? TABW_TW1 : TAB_TW1), o_tw2 = g.xwide ? TABX_TW2 : (g.wide ? TABW_TW2 : TAB_TW2);
    if (g.pair) { for (size_t l = 0; l < L; ++l) { tab[TAB_WIN + 2 * l] = win[l]; tab[TAB_WIN + 2 * l + 1] = win[l]; } }      // (w, w): one factor for frame A and B
    else if (g.xwide) { for (size_t l = 0; l < L && l < nfft; ++l) { tab[TABX_WIN + padx((int)l)] = win[l]; } }
    else { for (size_t l = 0; l < L && l < nfft; ++l) { tab[o_win + l] = win[l]; } }
    const long double pi = 3.141592653589793238462643383279502884L;
    for (int k2 = 0; k2 < 16; ++k2) { for (int n1 = 0; n1 < 16; ++n1) {
        const long double a = -2.0L * pi * (long double)(n1 * k2) / (long double)NC;
        tab[o_tw1 + (k2 * 16 + n1) * 2] = (float)cosl(a); tab[o_tw1 + (k2 * 16 + n1) * 2 + 1] = (float)sinl(a); } }
    for (int k1 = 0; k1 < 8; ++k1) { for (int k2 = 0; k2 < 16; ++k2) {
        const long double a = -pi * (long double)(k2 + 16 * k1) / (long double)NC;
        tab[o_tw2 + (k1 * 16 + k2) * 2] = (float)cosl(a); tab[o_tw2 + (k1 * 16 + k2) * 2 + 1] = (float)sinl(a); } }
    if (g.wide)
    {
        for (int k1 = 0; k1 < 8; ++k1) { for (int k2 = 0; k2 < 16; ++k2) {
            const long double a = -2.0L * pi * (long double)(k2 + 16 * k1) / 1024.0L;
            tab[TABW_TW3 + (k1 * 16 + k2) * 2] = (float)cosl(a); tab[TABW_TW3 + (k1 * 16 + k2) * 2 + 1] = (float)sinl(a); } }
    }
    if (g.xwide)
    {
        for (int k1 = 0; k1 < 9; ++k1) { for (int k2 = 0; k2 < 16; ++k2) {
            const long double a3 = -2.0L * pi * (long double)(k2 + 16 * k1) / 1024.0L, a4 = -2.0L * pi * (long double)(k2 + 16 * k1) / 2048.0L;
            tab[TABX_TW3 + (k1 * 16 + k2) * 2] = (float)cosl(a3); tab[TABX_TW3 + (k1 * 16 + k2) * 2 + 1] = (float)sinl(a3);
            tab[TABX_TW4 + (k1 * 16 + k2) * 2] = (float)cosl(a4); tab[TABX_TW4 + (k1 * 16 + k2) * 2 + 1] = (float)sinl(a4); } }
    }
    FastTables *t = new (std::nothrow) FastTables();
    if (!t) { if (err) { *err = "out of host memory"; } return nullptr; }
    t->d_tab = t->d_melw = t->d_mat2 = t->d_aux = nullptr;
    t->nz = g.nz; t->kind = o.kind; t->geom = g;
    t->B = t->C = t->P = t->MP = t->Cp = t->K2 = t->ntile_m = t->ntile_c = 0;
    t->F = (int)(nfft / 2u + 1u);
    t->D = t->F;
    const int klim = g.xwide ? PPX : (g.wide ? PPW : (g.pair ? PP2 : PP));                      // power-tile columns a k-loop may touch
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
        // 4 tiles (two row tiles each: a job per wave) up to 64 columns, else 8.
        // 2048-point plans have ONE row tile per mel tile (8 frames per workgroup) and four times the bins: a job per wave
        // takes 8 tiles (-2 %; no gain on the 1024-point plans)
        const int T = g.xwide ? std::min(8, Bi) : ((Bi <= 64) ? std::min(4, Bi) : 8);
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
        const int cap = g.xwide ? 100 * 1024 : 64 * 1024;
        e = opt_in_lds(plan_kernel(t, dither, false), cap);
        if (e == hipSuccess) { e = opt_in_lds(plan_kernel(t, dither, true), cap); }
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

hipError_t launch_fast(const FastTables *t, const FrameParams<float> &fp, hipStream_t s, hipEvent_t ev0, hipEvent_t ev1)
{
    if (fp.ntiles == 0) { return hipSuccess; }
    FastParams p;
    memset(&p, 0, sizeof(p));
    p.X = fp.X; p.X16 = fp.X16; p.Y = fp.Y; p.utts = fp.utts; p.tiles = fp.tiles; p.ntiles = fp.ntiles;
    p.L = fp.L; p.stp = fp.stp; p.D = fp.D;
    p.invL = 1.0f / (float)fp.L;
    const FastGeom &g = t->geom;
    p.span = (int)g.span; p.ws = (int)g.ws; p.m2 = (int)g.m2;
    p.snip_edges = fp.snip_edges; p.dc0 = fp.dc0; p.want_pre = fp.want_pre; p.want_post = fp.want_post;
    p.preemph = fp.do_preemph ? fp.preemph : 0.0f;
    p.do_dither = fp.do_dither; p.dither = fp.dither; p.seed = fp.seed;
#ifdef SPEECH_DIAG
    { const char *ab = getenv("SPEECH_ABLATE"); p.ablate = ab ? atoi(ab) : 0; }     // libspeech_diag.so only: timing diagnostics, DESIGN.md 4.3
#endif
    p.tab = t->d_tab;
    p.melw = t->d_melw; p.mat2 = t->d_mat2; p.aux = t->d_aux; p.plp_en = fp.plp_en;
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
    if (t->kind == SPEECH_PLP)                                              // second launch: autocorrelation rows -> cepstra, in place
    {
        if (fp.use_energy && !fp.plp_en) { return hipErrorInvalidValue; }
        PlpTailParams q;
        q.Y = fp.Y; q.utts = fp.utts; q.aux = t->d_aux; q.en = fp.plp_en; q.B = t->B; q.C = t->C; q.use_energy = fp.use_energy;
        e = launch_plp_tail(q, fp.nutt, fp.max_frames, s);
        if (e != hipSuccess) { return e; }
    }
    if (ev1) { e = hipEventRecord(ev1, s); }
    return e;
}

}  // namespace speech
