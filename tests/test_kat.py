"""Known-answer tests from closed forms -- independent of every implementation in this repository (oracle, numpy
statement, HIP kernels).  The reference holds no golden vectors, so these are the only absolute anchors
of the FFT indexing, its scaling, the bin-0 / last-bin quirks and the mel / DCT tables besides the float64 statement.

spectrogram (c/kaldi_spectrogram.c:236-251), rectangular window, no DC removal, no pre-emphasis, L = 400, nfft = 512:
  * constant c:        |X[k]|^2 = c^2 sin^2(pi 400 k / 512) / sin^2(pi k / 512),  bin 0 := frame energy 400 c^2
  * impulse A at l0:   |X[k]|^2 = A^2 for every k,                                bin 0 := A^2
  * complex-free tone: x[l] = A cos(2 pi m l / 512) over 400 samples has the Dirichlet-kernel closed form below
mfcc on a flat spectrum: all mel sums equal -> log-mel constant -> only cepstrum 0 is non-zero (DCT-II of a constant).
"""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(__file__))
import oracle_lib as O  # noqa: E402

L, NFFT, F = 400, 512, 257
FLAT = dict(dc0=0, preemph=0.0, win="rectangular", raw_energy=1, snip=1)
EPS = float(np.finfo(np.float32).eps)


def _dirichlet(k, n=L, nfft=NFFT):
    """|sum_{l<n} exp(-2 pi i k l / nfft)|^2"""
    k = np.asarray(k, dtype=np.float64)
    with np.errstate(divide="ignore", invalid="ignore"):
        v = np.sin(np.pi * n * k / nfft) ** 2 / np.sin(np.pi * k / nfft) ** 2
    return np.where(np.isclose(np.sin(np.pi * k / nfft), 0.0), float(n) ** 2, v)


def expected_constant(c):
    p = c * c * _dirichlet(np.arange(F))
    p[0] = L * c * c                      # bin 0 := raw energy
    p[F - 1] = p[F - 2]                   # last bin := the one before
    return np.log(np.maximum(p, EPS))


def expected_impulse(a):
    return np.full(F, np.log(a * a))


def expected_tone(a, m):
    """x[l] = a cos(2 pi m l / 512), l < 400: X[k] = a/2 (D(k-m) + D(k+m)) with D the complex Dirichlet sum"""
    k = np.arange(F, dtype=np.float64)

    def d(q):
        w = np.exp(-2j * np.pi * q / NFFT)
        with np.errstate(divide="ignore", invalid="ignore"):
            v = (1 - w ** L) / (1 - w)
        return np.where(np.isclose(w, 1.0), L + 0j, v)

    x = 0.5 * a * (d(k - m) + d(k + m))
    p = np.abs(x) ** 2
    l = np.arange(L)
    p[0] = float(np.sum((a * np.cos(2 * np.pi * m * l / NFFT)) ** 2))
    p[F - 1] = p[F - 2]
    return np.log(np.maximum(p, EPS))


def _signals():
    c = np.full(L + 160 * 3, 1234.0, dtype=np.float32)
    imp = np.zeros(L, dtype=np.float32)
    imp[37] = 2500.0
    l = np.arange(L)
    tone = (3000.0 * np.cos(2 * np.pi * 40 * l / NFFT)).astype(np.float32)
    return c, imp, tone


def _check(run_spectrogram, tol):
    c, imp, tone = _signals()
    y = run_spectrogram(c)
    assert y.shape == (4, F)
    want = expected_constant(1234.0)
    strong = want > want.max() - np.log(1e4)          # bins within 40 dB of the peak: the others sit at float32 round-off
    for w in range(4):
        assert np.abs(y[w] - want)[strong].max() < tol, w
    y = run_spectrogram(imp)
    assert np.abs(y[0] - expected_impulse(2500.0)).max() < tol
    y = run_spectrogram(tone)
    want = expected_tone(3000.0, 40)
    strong = want > want.max() - np.log(1e4)
    assert np.abs(y[0] - want)[strong].max() < tol
    assert int(np.argmax(y[0][1:F - 1])) + 1 == 40


def test_oracle_spectrogram_closed_forms():
    _check(lambda x: O.run("spectrogram", x, **FLAT)[1], 2e-5)
    _check(lambda x: O.run("spectrogram", x, single=False, **FLAT)[1], 1e-6)     # float64 oracle; float32 input values


def test_oracle_mfcc_of_a_flat_spectrum():
    """an impulse has a flat power spectrum A^2: every mel bin sums its weights times A^2; with lof/hif covering the whole
    band the sums of the triangular weights differ per bin, so use the property on the DCT instead: shifting the log-mel
    by a constant (scaling the impulse) moves only cepstrum 0"""
    a = np.zeros(L, dtype=np.float32); a[10] = 1000.0
    b = np.zeros(L, dtype=np.float32); b[10] = 4000.0
    o = dict(FLAT, B=23, C=13, Q=0.0)
    ya = O.run("mfcc", a, **o)[1][0]
    yb = O.run("mfcc", b, **o)[1][0]
    # log-mel shifts by 2 ln 4; DCT-II orthonormal row 0 = sqrt(1/B) -> c0 shifts by sqrt(B) * 2 ln 4; the rest stay put
    assert abs((yb[0] - ya[0]) - np.sqrt(23.0) * 2.0 * np.log(4.0)) < 2e-4
    assert np.abs(yb[1:] - ya[1:]).max() < 2e-4


@pytest.mark.gpu
def test_hip_spectrogram_closed_forms():
    from speech_amd import api
    kw = dict(dc0=0, preemph=0.0, win_type="rectangular", raw_energy=1, snip_edges=1)
    _check(lambda x: api.kaldi_spectrogram(x, **kw), 2e-5)                                   # tuned f32 kernel
    _check(lambda x: api.kaldi_spectrogram(x.astype(np.float64), **kw), 1e-6)                # f64 (generic kernel)


@pytest.mark.gpu
def test_hip_mfcc_scaling_moves_only_c0():
    from speech_amd import api
    a = np.zeros(L, dtype=np.float32); a[10] = 1000.0
    b = np.zeros(L, dtype=np.float32); b[10] = 4000.0
    kw = dict(dc0=0, preemph=0.0, win_type="rectangular", raw_energy=1, snip_edges=1, B=23, C=13, Q=0.0)
    ya, yb = api.kaldi_mfcc(a, **kw)[0], api.kaldi_mfcc(b, **kw)[0]
    assert abs((yb[0] - ya[0]) - np.sqrt(23.0) * 2.0 * np.log(4.0)) < 2e-4
    assert np.abs(yb[1:] - ya[1:]).max() < 2e-4
