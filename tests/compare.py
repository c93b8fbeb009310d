"""Error measures shared by the CPU and GPU tests.

Tolerances come from BASELINE.json's north star: max abs error <= 1e-4 on log-mel and cepstral features,
<= 1e-5 relative on the power spectrum.  The reference only ever emits LOG power (c/kaldi_spectrogram.c:251); a
relative power error eps is a log error eps.  A float32 FFT is accurate relative to the frame's LARGEST component,
not per bin (absolute amplitude error ~ eps32 * |X|peak), so the spectrogram measure gives full weight to bins within
40 dB of the frame peak and weights weaker bins by their amplitude ratio to that level:
    err = max_k |dlog P_k| * min(1, sqrt(P_k / (1e-4 P_peak)))
i.e. 1e-5 relative on the bins that carry the energy and 1e-7 |X|peak in amplitude below."""
import numpy as np


def max_err(kind, o, y, ref):
    """Scalar error of y against ref in the unit the tolerance is stated in."""
    y = np.asarray(y, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    assert y.shape == ref.shape, (y.shape, ref.shape)
    if not np.isfinite(y).all():
        return np.inf
    if kind == "spectrogram":
        if o.get("mn0", 0):
            return np.abs(y - ref).max()
        body = ref[:, 1:]                                   # column 0 is the frame log-energy, not a spectral bin
        w = np.minimum(1.0, np.exp(0.5 * (body - body.max(axis=1, keepdims=True))) / 1e-2)
        return max(np.abs(y[:, 0] - ref[:, 0]).max(), (np.abs(y[:, 1:] - body) * w).max())
    if kind == "fbank" and not o.get("lg", 1):
        # linear mel energies: relative to the frame's largest bin (first column may be a LOG energy: compare it in abs)
        if o.get("use_energy", 0):
            e0 = np.abs(y[:, 0] - ref[:, 0]).max()
            y, ref = y[:, 1:], ref[:, 1:]
        else:
            e0 = 0.0
        if o.get("mn0", 0):
            scale = np.abs(ref).max() + 1e-30
        else:
            scale = np.abs(ref).max(axis=1, keepdims=True) + 1e-30
        return max(e0, (np.abs(y - ref) / scale).max())
    return np.abs(y - ref).max()


# every check_parity call of a test session: (label, kind, clause that passed (1, 2) or 0, e_go, e_gt, e_ot); conftest.py
# prints how many needed clause 2 and tools/parity_table.py commits the same numbers per case
PARITY_LOG = []
MARGIN = 1.25          # clause 2: HIP may be at most this much further from float64 than the float32 CPU arithmetic is
CAP_FACTOR = 50.0      # clause 2 never admits HIP-vs-oracle above CAP_FACTOR x tol, however noisy the case


def check_parity(kind, o, got, oracle, truth=None, tol=1e-4, margin=MARGIN, oracle_fast=None, cap=None, label=None):
    """The acceptance rule of SURVEY.md 7.3-1.
    Clause 1: HIP-vs-oracle <= tol (the north star's tolerance).
    Clause 2, for cases that are ill-conditioned in float32 (the reference-style float32 CPU computation is itself further
    than tol from a float64 statement of the same algorithm): HIP-vs-truth <= max(tol, margin * oracle-vs-truth) AND
    HIP-vs-oracle <= cap (default CAP_FACTOR * tol), so that a weak bin cannot hide an arbitrarily wrong value.
    `oracle` is the exact-FFT oracle output, `oracle_fast` (optional) the same oracle with its FFT/DCT in float32 like
    FFTW's float codelets; the noise floor e_ot is the larger of the two distances to the truth.
    Returns (ok, e_go, e_gt, e_ot)."""
    e_go = max_err(kind, o, got, oracle)
    if truth is None:
        PARITY_LOG.append((label, kind, 1 if e_go <= tol else 0, e_go, None, None))
        return e_go <= tol, e_go, None, None
    e_gt = max_err(kind, o, got, truth)
    e_ot = max_err(kind, o, oracle, truth)
    if oracle_fast is not None:
        e_ot = max(e_ot, max_err(kind, o, oracle_fast, truth))
    cap = CAP_FACTOR * tol if cap is None else cap
    clause = 1 if e_go <= tol else (2 if (e_gt <= max(tol, margin * e_ot) and e_go <= cap) else 0)
    PARITY_LOG.append((label, kind, clause, e_go, e_gt, e_ot))
    return clause != 0, e_go, e_gt, e_ot


def float32_floor(kind, o, oracle, oracle_fast, truth=None):
    """How far two float32 CPU evaluations of the reference arithmetic are from each other on this input (exact-FFT oracle vs
    float-FFT oracle), and, when a float64 statement is given, from it: the reference's own float32 noise, the floor under
    which no float32 implementation can be told from another."""
    f = max_err(kind, o, oracle_fast, oracle)
    if truth is not None:
        f = max(f, max_err(kind, o, oracle, truth), max_err(kind, o, oracle_fast, truth))
    return f
