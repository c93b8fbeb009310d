"""CPU oracle of the corpus post-processing passes.  TEST INFRASTRUCTURE ONLY: imported by tests/,
never by the product path.

PARITY UNPINNED: the reference has only mn0 (c/kaldi_fbank.c:355-370), and that with a stride defect (SURVEY.md 8(a)
row N); CMVN, deltas and splicing restate the published behaviour of Kaldi's apply-cmvn, add-deltas (DeltaFeatures in
feature-functions.cc), splice-feats and subsample-feats, which are not in /root/reference.  Plain numpy, one utterance (W x D) at a time.
"""
import numpy as np


def cmvn(y, norm_vars=False):
    """per-utterance: x * scale + offset with scale = 1/std (floored variance 1e-20) or 1, offset = -mean * scale; the
    statistics in float64, the two steps rounded in y's dtype as MulColsVec / AddVecToRows do"""
    y64 = y.astype(np.float64)
    n = float(y.shape[0])
    mean = y64.sum(axis=0) / n
    scale = np.ones_like(mean)
    if norm_vars:
        scale = 1.0 / np.sqrt(np.maximum((y64 * y64).sum(axis=0) / n - mean * mean, 1.0e-20))
    offset = -mean * scale
    return (y * scale.astype(y.dtype)).astype(y.dtype) + offset.astype(y.dtype)


def affine(y, scale, offset):
    return (y * np.asarray(scale).astype(y.dtype)).astype(y.dtype) + np.asarray(offset).astype(y.dtype)


def delta_filters(order, window):
    """Kaldi DeltaFeatures: scales[0] = [1]; scales[k] = scales[k-1] (*) [j / sum j^2 for j = -window..window]"""
    scales = [np.array([1.0])]
    for k in range(1, order + 1):
        prev = scales[-1]
        ph, ch = (len(prev) - 1) // 2, (len(prev) - 1) // 2 + window
        cur = np.zeros(len(prev) + 2 * window)
        norm = 0.0
        for j in range(-window, window + 1):
            norm += j * j
            for i in range(-ph, ph + 1):
                cur[j + i + ch] += j * prev[i + ph]
        scales.append(cur / norm)
    return scales


def add_deltas(y, order=2, window=2):
    """[y | delta | delta-delta ...], each filter applied to y with the frame index clamped to [0, W-1]; float64 sums"""
    W, D = y.shape
    out = np.zeros((W, D * (order + 1)))
    for k, sc in enumerate(delta_filters(order, window)):
        half = (len(sc) - 1) // 2
        for j in range(-half, half + 1):
            idx = np.clip(np.arange(W) + j, 0, W - 1)
            out[:, k * D:(k + 1) * D] += sc[j + half] * y[idx].astype(np.float64)
    return out


def splice(y, left, right):
    W, D = y.shape
    return np.concatenate([y[np.clip(np.arange(W) + c, 0, W - 1)] for c in range(-left, right + 1)], axis=1)


def subsample(y, n, offset=0):
    """Kaldi subsample-feats --n=n --offset=offset (n > 0): frames offset, offset + n, ..."""
    return y[offset::n]
