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
MARGIN = 1.25          # clause 2: HIP's rounding-noise LEVEL (rms distance to float64) may be at most this multiple of the float32 CPU arithmetic's
MAX_MARGIN = 2.0       # clause 2: and its largest single deviation from float64 at most this multiple of the CPU arithmetic's largest
CAP_FACTOR = 50.0      # clause 2 never admits HIP-vs-oracle above CAP_FACTOR x tol, however noisy the case


def check_parity(kind, o, got, oracle, truth=None, tol=1e-4, margin=MARGIN, oracle_fast=None, cap=None, label=None, max_margin=MAX_MARGIN):
    """The acceptance rule of SURVEY.md 7.3-1.
    Clause 1: HIP-vs-oracle <= tol (the north star's tolerance), in the maximum norm.
    Clause 2, for cases that are ill-conditioned in float32 (the reference-style float32 CPU computation is itself further
    than tol from a float64 statement of the same algorithm), all of:
      (a) rms(HIP - truth) <= margin * rms(oracle - truth): the same noise level as the reference arithmetic.  The margin is
          on the rms because the MAXIMUM of ~1e5 rounding errors scatters by +-50 % between two equally accurate float32
          implementations (measured: profiles/r2_parity_table.md, e_gt / e_ot between 0.63 and 1.98 over 230 cases);
      (b) max|HIP - truth| <= max(tol, max_margin * max|oracle - truth|): no single value far outside that noise;
      (c) max|HIP - oracle| <= cap (default CAP_FACTOR * tol): an absolute bound, however noisy the case.
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
    if oracle_fast is not None:
        e_ot = max(e_ot, max_err(kind, o, oracle_fast, truth))
        r_ot = max(r_ot, rms_err(kind, o, oracle_fast, truth))
    cap = case_cap(kind, o, tol) if cap is None else cap
    ratio = r_gt / r_ot if r_ot > 0 else (0.0 if r_gt == 0 else np.inf)
    ok2 = (ratio <= margin or e_gt <= tol) and e_gt <= max(tol, max_margin * e_ot) and e_go <= cap
    clause = 1 if e_go <= tol else (2 if ok2 else 0)
    PARITY_LOG.append((label, kind, clause, e_go, e_gt, e_ot, ratio))
    return clause != 0, e_go, e_gt, e_ot


def case_cap(kind, o, tol):
    """The absolute cap of clause 2 for a case.  One family is exempt: PLP without spectral compression (compress = 1.0).
    Levinson-Durbin of order 12 on the raw mel power spectrum (60 dB of dynamic range) is ill-conditioned in float32 whatever
    the input: the reference-style CPU arithmetic with an exact and with a float FFT already differ by 5e-3 .. 6e-2 on it
    (tests/cases.py seed 520; tools/parity_table.py), so only clauses 2(a) and 2(b) can speak there."""
    if kind == "plp" and float(o.get("compress", 1.0 / 3.0)) == 1.0:
        return np.inf
    return CAP_FACTOR * tol


def float32_floor(kind, o, oracle, oracle_fast, truth=None):
    """How far two float32 CPU evaluations of the reference arithmetic are from each other on this input (exact-FFT oracle vs
    float-FFT oracle), and, when a float64 statement is given, from it: the reference's own float32 noise, the floor under
    which no float32 implementation can be told from another."""
    f = max_err(kind, o, oracle_fast, oracle)
    if truth is not None:
        f = max(f, max_err(kind, o, oracle, truth), max_err(kind, o, oracle_fast, truth))
    return f
