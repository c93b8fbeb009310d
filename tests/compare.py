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


def check_parity(kind, o, got, oracle, truth=None, tol=1e-4, margin=2.0, oracle_fast=None):
    """The acceptance rule of SURVEY.md 7.3-1: HIP-vs-oracle <= tol, OR (where the reference-style float32 CPU
    computation is itself further than tol/margin from a float64 statement of the same algorithm, i.e. the case is
    ill-conditioned in float32) HIP-vs-truth <= margin * oracle-vs-truth.  `oracle` is the exact-FFT oracle output,
    `oracle_fast` (optional) the same oracle with its FFT/DCT in float32 like FFTW's float codelets; the noise floor
    e_ot is the larger of the two distances to the truth.  Returns (ok, e_go, e_gt, e_ot)."""
    e_go = max_err(kind, o, got, oracle)
    if truth is None:
        return e_go <= tol, e_go, None, None
    e_gt = max_err(kind, o, got, truth)
    e_ot = max_err(kind, o, oracle, truth)
    if oracle_fast is not None:
        e_ot = max(e_ot, max_err(kind, o, oracle_fast, truth))
    ok = (e_go <= tol) or (e_gt <= max(tol, margin * e_ot))
    return ok, e_go, e_gt, e_ot
