"""Error measures shared by the CPU and GPU tests.

Tolerances come from BASELINE.json's north star: max abs error <= 1e-4 on log-mel and cepstral features,
<= 1e-5 relative on the power spectrum.  The reference only ever emits LOG power (c/kaldi_spectrogram.c:251); a
relative power error eps is a log error eps.  A float32 FFT is accurate relative to the frame's LARGEST component,
not per bin (absolute amplitude error ~ eps32 * |X|peak), so the spectrogram measure gives full weight to bins within
40 dB of the frame peak and weights weaker bins by their amplitude ratio to that level:
    err = max_k |dlog P_k| * min(1, sqrt(P_k / (1e-4 P_peak)))
i.e. 1e-5 relative on the bins that carry the energy and 1e-7 |X|peak in amplitude below."""
import numpy as np


def rms_err(kind, o, y, ref):
    """Root-mean-square of the same weighted differences whose maximum max_err returns: the LEVEL of the rounding noise, which
    -- unlike the maximum of a few hundred thousand noisy values -- does not scatter from one implementation to the next."""
    return max_err(kind, o, y, ref, stat=lambda a: float(np.sqrt(np.mean(np.square(a)))))


def max_err(kind, o, y, ref, stat=None):
    """Scalar error of y against ref in the unit the tolerance is stated in."""
    stat = stat or (lambda a: float(np.max(a)))
    y = np.asarray(y, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    assert y.shape == ref.shape, (y.shape, ref.shape)
    if not np.isfinite(y).all():
        return np.inf
    if kind == "spectrogram":
        if o.get("mn0", 0):
            return stat(np.abs(y - ref))
        body = ref[:, 1:]                                   # column 0 is the frame log-energy, not a spectral bin
        w = np.minimum(1.0, np.exp(0.5 * (body - body.max(axis=1, keepdims=True))) / 1e-2)
        return max(stat(np.abs(y[:, 0] - ref[:, 0])), stat(np.abs(y[:, 1:] - body) * w))
    if kind == "fbank" and not o.get("lg", 1):
        # linear mel energies: relative to the frame's largest bin (first column may be a LOG energy: compare it in abs)
        if o.get("use_energy", 0):
            e0 = stat(np.abs(y[:, 0] - ref[:, 0]))
            y, ref = y[:, 1:], ref[:, 1:]
        else:
            e0 = 0.0
        if o.get("mn0", 0):
            scale = np.abs(ref).max() + 1e-30
        else:
            scale = np.abs(ref).max(axis=1, keepdims=True) + 1e-30
        return max(e0, stat(np.abs(y - ref) / scale))
    return stat(np.abs(y - ref))


# every check_parity call of a test session: (label, kind, clause that passed (1, 2) or 0, e_go, e_gt, e_ot, rms ratio); conftest.py
# prints how many needed clause 2 and tools/parity_table.py commits the same numbers per case
PARITY_LOG = []
MARGIN = 1.25          # clause 2(a): HIP's rounding-noise LEVEL (rms distance to float64) may be at most this multiple of the float32 CPU arithmetic's
# clause 2(b): its largest single deviation from float64 at most this multiple of the CPU arithmetic's largest.  The maximum of
# ~1e5 rounding errors of two equally accurate float32 implementations scatters: measured distribution of e_gt / e_ot in
# profiles/r3_parity_table.md (every case of cases.py: 0.31 .. 2.10, median 0.98, 99th percentile 1.8) and over the GPU suite's
# ~590 checks incl. the random option sets of tests/test_fuzz.py (largest 2.10: cases.py seed 942, rms noise ratio 0.97); ONE
# bound for every case and every test -- no per-case seeds, lengths or margins (round 2 had 2.0 here, 3.0 in the fuzz test for
# 1024/2048-point frames, and two re-drawn cases)
MAX_MARGIN = 2.5
# clause 2(c): absolute cap on HIP-vs-oracle, per kind, in units of the tolerance -- about 1.3 x the largest value the GPU suite
# and profiles/r3_parity_table.md hold for the kind (round 2 had 50 x for everything): fbank 18.5 x (80 bins on tonal audio: the
# lowest filters are single FFT bins), mfcc 15.9 x (128 mel bins / 80 cepstra in test_tuned_and_generic_kernels_agree, where the
# CPU arithmetic itself is 65 x tol from float64; 9.2 x otherwise), plp 12.7 x (2048-point frames, compress = 0.5, CPU
# arithmetic 27 x tol from float64), spectrogram 2.9 x
CAP_FACTORS = {"fbank": 25.0, "mfcc": 20.0, "spectrogram": 10.0, "plp": 15.0}
# plp with compress = 1.0 (no spectral compression): order-12 Levinson-Durbin on the raw mel power spectrum (60 dB of dynamic
# range) is ill-conditioned in float32 whatever the input -- the reference-style CPU arithmetic with an exact and with a float
# FFT differ by e_ff = 5e-3 .. 6e-2 from each other there.  Its cap is tied to that measured distance on the same input
PLP_C1_CAP_FF = 3.0


def check_parity(kind, o, got, oracle, truth=None, tol=1e-4, margin=MARGIN, oracle_fast=None, cap=None, label=None, max_margin=MAX_MARGIN):
    """The acceptance rule of SURVEY.md 7.3-1.
    Clause 1: HIP-vs-oracle <= tol (the north star's tolerance), in the maximum norm.
    Clause 2, for cases that are ill-conditioned in float32 (the reference-style float32 CPU computation is itself further
    than tol from a float64 statement of the same algorithm), ALL of:
      (a) rms(HIP - truth) <= margin * rms(oracle - truth): the same noise level as the reference arithmetic.  The margin is
          on the rms because the MAXIMUM of ~1e5 rounding errors scatters between two equally accurate float32
          implementations (see MAX_MARGIN);
      (b) max|HIP - truth| <= max(tol, max_margin * max|oracle - truth|): no single value far outside that noise;
      (c) max|HIP - oracle| <= cap: an absolute bound per kind (case_cap), however noisy the case.
    (Round 2's rule let (a) pass whenever max|HIP - truth| <= tol; that bypass is gone: no case needs it.)
    `oracle` is the exact-FFT oracle output, `oracle_fast` (optional) the same oracle with its FFT/DCT in float32 like
    FFTW's float codelets; the noise floors are the larger of the two distances to the truth.
    Returns (ok, e_go, e_gt, e_ot)."""
    e_go = max_err(kind, o, got, oracle)
    if truth is None:
        PARITY_LOG.append((label, kind, 1 if e_go <= tol else 0, e_go, None, None, None))
        return e_go <= tol, e_go, None, None
    e_gt = max_err(kind, o, got, truth)
    e_ot = max_err(kind, o, oracle, truth)
    r_gt = rms_err(kind, o, got, truth)
    r_ot = rms_err(kind, o, oracle, truth)
    e_ff = None
    if oracle_fast is not None:
        e_ot = max(e_ot, max_err(kind, o, oracle_fast, truth))
        r_ot = max(r_ot, rms_err(kind, o, oracle_fast, truth))
        e_ff = max_err(kind, o, oracle_fast, oracle)
    cap = case_cap(kind, o, tol, e_ff) if cap is None else cap
    ratio = r_gt / r_ot if r_ot > 0 else (0.0 if r_gt == 0 else np.inf)
    ok2 = ratio <= margin and e_gt <= max(tol, max_margin * e_ot) and e_go <= cap
    clause = 1 if e_go <= tol else (2 if ok2 else 0)
    PARITY_LOG.append((label, kind, clause, e_go, e_gt, e_ot, ratio))
    return clause != 0, e_go, e_gt, e_ot


# Clause 2(a) compares two ESTIMATES of a noise level.  Over a handful of frames the estimate itself scatters: the recursions of
# plp (and a lifter gain of 12) put most of the squared error of a 10-frame input into one or two ill-conditioned frames, and
# which frames those are differs between float32 implementations (tools/plp_probe.py: on cases.py seed 521, N = 2000, the
# tuned AND the generic kernel measure 1.53 / 1.44 x the CPU arithmetic, on seeds 522-524 of the same option set 0.86-1.09, and
# the CPU arithmetic's own two FFT modes differ per frame by up to 5 x).  So the level is estimated on at least
# POOL_MIN_FRAMES frames: a shorter case is run on POOL_SEEDS - 1 further signals of the same length and options and the rule
# is applied to all of them together (every clause, the case's own signal included) -- more data, not a wider margin.
POOL_MIN_FRAMES = 64
POOL_SEEDS = 8


def pool(arrays):
    """frames of several signals under one option set, one after the other"""
    return np.concatenate([np.asarray(a) for a in arrays], axis=0)


def case_cap(kind, o, tol, e_ff=None):
    """The absolute cap of clause 2(c) for a case: CAP_FACTORS[kind] x tol; plp without spectral compression (compress = 1.0):
    that, or PLP_C1_CAP_FF x the distance between the CPU arithmetic's own two FFT modes on this input where that is larger."""
    cap = CAP_FACTORS.get(kind, 10.0) * tol
    if kind == "plp" and float(o.get("compress", 1.0 / 3.0)) == 1.0 and e_ff is not None:
        cap = max(cap, PLP_C1_CAP_FF * e_ff)
    return cap


def float32_floor(kind, o, oracle, oracle_fast, truth=None):
    """How far two float32 CPU evaluations of the reference arithmetic are from each other on this input (exact-FFT oracle vs
    float-FFT oracle), and, when a float64 statement is given, from it: the reference's own float32 noise, the floor under
    which no float32 implementation can be told from another."""
    f = max_err(kind, o, oracle_fast, oracle)
    if truth is not None:
        f = max(f, max_err(kind, o, oracle, truth), max_err(kind, o, oracle_fast, truth))
    return f
