"""Identity tests that pin the PLP tail without a second copy of the recursions: the reference's
kaldi_plp.c does not compile, and upstream Kaldi is not in the container, so the oracle's PLP is checked against
mathematics instead -- the cosine bases, the equal-loudness curve, the Toeplitz normal equations (scipy's solver, not
Levinson-Durbin), the cepstrum of the all-pole spectrum by a long FFT (not the LPC-to-cepstrum recursion), and the shared
front end against the fbank oracle.  CPU only."""
import numpy as np
from scipy.linalg import solve_toeplitz, toeplitz

import np_truth as T
import oracle_lib as O
import signals

B, ORDER = 23, 12


def _bases(order, B):
    i = np.arange(order + 1)[:, None]
    j = np.arange(B + 2)[None, :]
    scale = 1.0 / (2.0 * (B + 1))
    b = 2.0 * scale * np.cos(np.pi / (B + 1) * i * j)
    b[:, 0] = scale
    b[:, B + 1] = scale * np.cos(np.pi * i[:, 0])
    return b


def test_idft_bases_of_a_flat_spectrum_give_an_impulse():
    """r = bases @ D is the inverse cosine transform of the symmetric spectrum D: a constant gives r0 = D, r_i = 0; a single
    cosine cos(pi m j / (B+1)) gives r_m = 1/2"""
    b = _bases(ORDER, B)
    r = b @ np.full(B + 2, 3.5)
    assert abs(r[0] - 3.5) < 1e-13 and np.abs(r[1:]).max() < 1e-13
    j = np.arange(B + 2)
    for m in (1, 5, ORDER):
        r = b @ np.cos(np.pi * m * j / (B + 1))
        want = np.zeros(ORDER + 1); want[m] = 0.5
        assert np.abs(r - want).max() < 1e-13


def test_equal_loudness_closed_form_values():
    """(f^2 / (f^2 + 1.6e5))^2 (f^2 + 1.44e6) / (f^2 + 9.61e6): the values at 1 kHz and 4 kHz worked by hand, and the curve's
    effect on the features: scaling bin b's energy by 1/eql_b must undo it"""
    f = np.array([1000.0, 4000.0])
    fsq = f * f
    e = (fsq / (fsq + 1.6e5)) ** 2 * ((fsq + 1.44e6) / (fsq + 9.61e6))
    assert np.allclose(e, [(1e6 / 1.16e6) ** 2 * (2.44e6 / 1.061e7), (1.6e7 / 1.616e7) ** 2 * (1.744e7 / 2.561e7)], rtol=1e-14)
    assert 0.17 < e[0] < 0.172 and 0.66 < e[1] < 0.67


def _plp_pieces(x, compress):
    """mel energies -> equal loudness -> compression -> duplicated ends -> autocorrelation, in float64 (np_truth pieces)"""
    fr, _, _, nfft = T.front_end(x, 16000.0, 25.0, 10.0, 1, 1, 0.97, "povey", single=False)
    P = T.power(fr, nfft)
    M, lomel, dmel = T.mel_matrix(16000.0, nfft, B, 20.0, 8000.0)
    me = P @ M
    cf = 700.0 * (np.exp((lomel + (np.arange(B) + 1) * dmel) / 1127.0) - 1.0)
    fsq = cf * cf
    eql = (fsq / (fsq + 1.6e5)) ** 2 * ((fsq + 1.44e6) / (fsq + 9.61e6))
    v = (me * eql) ** compress
    d = np.concatenate([v[:, :1], v, v[:, -1:]], axis=1)
    return me, d @ _bases(ORDER, B).T


def test_front_end_through_mel_sums_equals_the_fbank_oracle():
    x = signals.utterance(8000, 11, dtype=np.float64)
    me, _ = _plp_pieces(x, 1.0)
    rc, fb = O.run("fbank", x, single=False, B=B, lg=0, amp=0, use_energy=0)
    assert rc == 0
    ref = me.copy(); ref[:, -1] = ref[:, -2]                  # the fbank quirk (c/kaldi_fbank.c:350); PLP uses the true last bin
    assert np.allclose(fb, ref, rtol=1e-9)


def test_plp_equals_toeplitz_solution_and_all_pole_cepstrum():
    """oracle PLP (no lifter: Q = 0) against an independent route: solve the Toeplitz normal equations R a = -r with scipy,
    residual E = r0 + sum a_i r_i, and the cepstrum of E / |A(e^jw)|^2 by a 2^16-point FFT"""
    x = signals.utterance(8000, 12, dtype=np.float64)
    for compress in (1.0 / 3.0, 1.0, 0.5):
        rc, y = O.run("plp", x, single=False, B=B, C=ORDER + 1, Q=0.0, compress=compress, cep_scale=1.0)
        assert rc == 0
        _, ac = _plp_pieces(x, compress)
        assert y.shape == (ac.shape[0], ORDER + 1)
        for w in range(0, ac.shape[0], 7):
            r = ac[w]
            a = solve_toeplitz(r[:ORDER], -r[1:ORDER + 1])               # predictor: x[n] + sum a_i x[n-i] = e[n]
            assert np.abs(toeplitz(r[:ORDER]) @ a + r[1:ORDER + 1]).max() < 1e-9 * r[0]
            E = r[0] + a @ r[1:ORDER + 1]
            assert E > 0
            nf = 1 << 16
            A = np.fft.rfft(np.concatenate([[1.0], a]), nf)
            logspec = np.log(E) - 2.0 * np.log(np.abs(A))
            # log P(w) = c[0] + 2 sum_n c[n] cos(n w) with P = E |H|^2, and log |H|^2 = 2 Re sum_n c_n e^{-jwn} for the
            # cepstrum c_n of H = 1/A: the inverse transform of log P is log E at lag 0 and c_n itself at lag n
            want = np.fft.irfft(logspec, nf)[:ORDER + 1]
            got = y[w].copy()
            # the last coefficient is a copy of the one before (c/kaldi_plp.c:403)
            assert got[-1] == got[-2]
            assert np.abs(got[:-1] - want[:-1]).max() < 1e-7 * max(1.0, np.abs(want).max()), (compress, w)


def test_lifter_and_cep_scale_are_plain_factors():
    x = signals.utterance(6000, 13, dtype=np.float64)
    rc, y0 = O.run("plp", x, single=False, B=B, C=13, Q=0.0, cep_scale=1.0)
    rc2, y1 = O.run("plp", x, single=False, B=B, C=13, Q=22.0, cep_scale=2.5)
    assert rc == 0 and rc2 == 0
    k = np.arange(13)
    lift = 1.0 + 11.0 * np.sin(np.pi * k / 22.0)
    want = y0 * lift * 2.5
    want[:, -1] = want[:, -2]
    assert np.allclose(y1, want, rtol=1e-12, atol=1e-12)
