"""The oracle against code that neither the reference's author nor this repository's wrote.

DESIGN.md 3: the reference holds no fixtures and cannot be built here, so parity with it is unpinned, and the oracle, the numpy
statement and the product's tables come from one reading of the reference.  What CAN be checked in this image: the reference
states that it computes Kaldi's features, and Hugging Face `transformers` (installed; `transformers.audio_utils`) carries a numpy
implementation of Kaldi's `compute-fbank-feats` front end -- Povey window, DC removal, pre-emphasis with Kaldi's first-sample rule,
power spectrum, triangular filters on Kaldi's mel scale (`mel_scale="kaldi", triangularize_in_mel_space=True`), log with the
FLT_EPSILON floor -- written to reproduce `torchaudio.compliance.kaldi.fbank` (it is what `ASTFeatureExtractor` and
`SeamlessM4TFeatureExtractor` fall back to without torchaudio).  scipy's orthonormal DCT-II is Kaldi's MFCC transform.

These tests hold the C oracle (double precision, i.e. the `_d` restatement) to that code: windows and mel weights to rounding,
the whole fbank / spectrogram pipelines to 3e-6 (`transformers` stores its spectrum as complex64; mfcc: that rounding through the DCT
and the lifter, and the DCT + lifter stage alone to 1e-9), on the elements where the
reference means Kaldi's value.  The reference's three documented quirks are the exceptions and are checked as such: the last
filterbank bin and the last cepstrum repeat their neighbour (c/kaldi_fbank.c:350, c/kaldi_mfcc.c:333), the spectrogram's bin 0
is the log energy and its Nyquist bin repeats bin N/2 - 1 (c/kaldi_spectrogram.c:236-251).
(`ASTFeatureExtractor` itself spaces its 256 bins over [0, sr/2] inclusive, 31.37 instead of 31.25 Hz apart; the filter bank here
is asked for the nfft/2 + 1 exact bin frequencies.)"""
import warnings

import numpy as np
import pytest

import np_truth as T
import oracle_lib as O
import signals

au = pytest.importorskip("transformers.audio_utils")
scipy_fft = pytest.importorskip("scipy.fft")


def _kaldi_mel(sr, nfft, B, lof, hif):
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        return au.mel_filter_bank(num_frequency_bins=nfft // 2 + 1, num_mel_filters=B, min_frequency=lof, max_frequency=hif, sampling_rate=int(sr),
                                  norm=None, mel_scale="kaldi", triangularize_in_mel_space=True)


def _kaldi_logspec(x, sr, fl, fs, mel=None, preemph=0.97, dc0=True, win="povey", power=2.0, log=True):
    L, stp = int(round(sr * fl / 1000.0)), int(round(sr * fs / 1000.0))
    nfft = 1 << (L - 1).bit_length()
    w = au.window_function(L, win, periodic=False)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        return au.spectrogram(np.asarray(x, dtype=np.float64), w, frame_length=L, hop_length=stp, fft_length=nfft, power=power, center=False,
                              preemphasis=preemph if preemph else None, mel_filters=mel, log_mel="log" if log else None, mel_floor=float(np.finfo(np.float32).eps),
                              remove_dc_offset=bool(dc0)).T, nfft


@pytest.mark.parametrize("name", ["povey", "hamming", "hann"])
@pytest.mark.parametrize("L", [200, 400, 551, 1200])
def test_windows_are_kaldis(name, L):
    np.testing.assert_allclose(T.window(L, name), au.window_function(L, name, periodic=False), rtol=0, atol=1e-15)


@pytest.mark.parametrize("sr,nfft,B,lof,hif", [(16000.0, 512, 23, 20.0, 8000.0), (16000.0, 512, 40, 20.0, 7600.0), (16000.0, 512, 80, 20.0, 8000.0),
                                               (8000.0, 256, 23, 20.0, 4000.0), (8000.0, 256, 15, 100.0, 3700.0), (22050.0, 1024, 80, 20.0, 11025.0),
                                               (48000.0, 2048, 80, 20.0, 24000.0), (16000.0, 512, 128, 0.0, 8000.0), (44100.0, 2048, 40, 60.0, 20000.0)])
def test_mel_weights_are_kaldis(sr, nfft, B, lof, hif):
    """every weight of every filter, the last filter included (the reference's quirk is in the features, not in the table)"""
    M, _, _ = T.mel_matrix(sr, nfft, B, lof, hif)
    K = _kaldi_mel(sr, nfft, B, lof, hif)
    assert M.shape == K.shape
    np.testing.assert_allclose(M, K, rtol=0, atol=1e-12)


CASES = [dict(sr=16000.0, fl=25.0, fs=10.0, B=80, lof=20.0, hif=8000.0),                  # BASELINE.json configs[2], the headline
         dict(sr=16000.0, fl=25.0, fs=10.0, B=23, lof=20.0, hif=8000.0),
         dict(sr=16000.0, fl=25.0, fs=10.0, B=40, lof=20.0, hif=7600.0),
         dict(sr=8000.0, fl=25.0, fs=10.0, B=23, lof=20.0, hif=4000.0),
         dict(sr=22050.0, fl=25.0, fs=10.0, B=80, lof=20.0, hif=11025.0),
         dict(sr=16000.0, fl=20.0, fs=12.5, B=30, lof=100.0, hif=7000.0, preemph=0.0, dc0=0, win="hamming")]
IDS = ["-".join("%s=%s" % kv for kv in sorted(c.items())) for c in CASES]


def _audio(c, secs=2.0, seed=11):
    return signals.utterance(int(c["sr"] * secs), seed, sr=c["sr"], dtype=np.float64)


@pytest.mark.parametrize("c", CASES, ids=IDS)
def test_oracle_fbank_is_kaldis_fbank_but_for_the_last_bin(c):
    x = _audio(c)
    o = dict(c)
    rc, y = O.run("fbank", x, single=False, **o)
    assert rc == 0
    K, nfft = _kaldi_logspec(x, c["sr"], c["fl"], c["fs"], mel=_kaldi_mel(c["sr"], 1 << (int(round(c["sr"] * c["fl"] / 1000.0)) - 1).bit_length(), c["B"], c["lof"], c["hif"]),
                             preemph=c.get("preemph", 0.97), dc0=c.get("dc0", 1), win=c.get("win", "povey"))
    B = c["B"]
    assert y.shape == K.shape
    assert np.abs(y[:, :B - 1] - K[:, :B - 1]).max() < 3e-6                  # Kaldi's value in every bin but the last
    assert np.array_equal(y[:, B - 1], y[:, B - 2])                          # the last bin repeats its neighbour (c/kaldi_fbank.c:350) ...
    assert np.abs(y[:, B - 1] - K[:, B - 1]).max() > 1e-3                    # ... which is not Kaldi's last bin


@pytest.mark.parametrize("amp,lg", [(1, 1), (0, 0), (1, 0)])
def test_oracle_fbank_on_magnitudes_and_without_the_log(amp, lg):
    """the reference's two fbank switches (c/kaldi_fbank.c:312-316, :346): filters on |X| instead of |X|^2, and no log"""
    c = CASES[1]
    x = _audio(c, seed=14)
    rc, y = O.run("fbank", x, single=False, amp=amp, lg=lg, **c)
    assert rc == 0
    K, _ = _kaldi_logspec(x, c["sr"], c["fl"], c["fs"], mel=_kaldi_mel(c["sr"], 512, c["B"], c["lof"], c["hif"]), power=1.0 if amp else 2.0, log=bool(lg))
    B = c["B"]
    err = np.abs(y[:, :B - 1] - K[:, :B - 1]) / (1.0 if lg else np.maximum(np.abs(K[:, :B - 1]), 1e-30))
    assert err.max() < 3e-6
    assert np.array_equal(y[:, B - 1], y[:, B - 2])


@pytest.mark.parametrize("c", CASES[:5], ids=IDS[:5])
def test_oracle_spectrogram_is_kaldis_log_power_spectrum_between_its_two_quirks(c):
    x = _audio(c, seed=12)
    rc, y = O.run("spectrogram", x, single=False, sr=c["sr"], fl=c["fl"], fs=c["fs"])
    assert rc == 0
    K, nfft = _kaldi_logspec(x, c["sr"], c["fl"], c["fs"])
    assert y.shape == K.shape == (y.shape[0], nfft // 2 + 1)
    # transformers keeps the spectrum as complex64: a bin 90 dB under the frame's peak carries that rounding; compare where the
    # north star's tolerance looks (within 40 dB of the peak)
    strong = K[:, 1:-1] > K[:, 1:-1].max(axis=1, keepdims=True) - 9.2
    assert np.abs(y[:, 1:-1] - K[:, 1:-1])[strong].max() < 3e-6
    assert np.array_equal(y[:, -1], y[:, -2])                                # Nyquist bin := bin N/2 - 1


@pytest.mark.parametrize("c", CASES[1:4], ids=IDS[1:4])
@pytest.mark.parametrize("C,Q", [(13, 22.0), (None, 22.0), (8, 0.0)])
def test_oracle_mfcc_is_kaldis_mfcc_but_for_the_last_cepstrum(c, C, Q):
    """Kaldi's MFCC = its lifter times the orthonormal DCT-II (scipy) of its log-mel energies (transformers)"""
    x = _audio(c, seed=13)
    B = c["B"]
    C = B if C is None else C
    rc, y = O.run("mfcc", x, single=False, C=C, Q=Q, **c)
    assert rc == 0
    K, nfft = _kaldi_logspec(x, c["sr"], c["fl"], c["fs"], mel=_kaldi_mel(c["sr"], 1 << (int(round(c["sr"] * c["fl"] / 1000.0)) - 1).bit_length(), B, c["lof"], c["hif"]))
    k = np.arange(C)
    lift = 1.0 + 0.5 * Q * np.sin(np.pi * k / Q) if Q > 0 else np.ones(C)
    want = scipy_fft.dct(K, type=2, norm="ortho", axis=1)[:, :C] * lift[None, :]
    assert y.shape == want.shape
    # (1) the whole chain against the third-party log-mels: a cepstrum sums B of them, each carrying transformers' complex64 rounding
    # (<= 3e-6, the fbank test above), with weights sqrt(2 / B) cos(..), and the lifter multiplies by up to 1 + Q / 2
    assert np.abs(y[:, :C - 1] - want[:, :C - 1]).max() < 3e-6 * np.sqrt(2.0 * B) * lift.max()
    # (2) the DCT + lifter stage alone at full precision: scipy's orthonormal DCT-II of the float64 statement's own log-mels (which
    # (1) and the fbank test tie to the third-party ones)
    fr, _, _, nfft = T.front_end(x, c["sr"], c["fl"], c["fs"], 1, 1, 0.97, "povey", False, energy_before_dc=True)
    lm = np.log(np.maximum(T.power(fr, nfft) @ T.mel_matrix(c["sr"], nfft, B, c["lof"], c["hif"])[0], T.FLT_EPS))
    exact = scipy_fft.dct(lm, type=2, norm="ortho", axis=1)[:, :C] * lift[None, :]
    assert np.abs(y[:, :C - 1] - exact[:, :C - 1]).max() < 1e-9
    assert np.array_equal(y[:, C - 1], y[:, C - 2])                          # c/kaldi_mfcc.c:333


@pytest.mark.gpu
@pytest.mark.parametrize("c", CASES[:5], ids=IDS[:5])
def test_hip_fbank_is_kaldis_fbank_to_the_float32_noise_of_the_reference_arithmetic(c):
    """the PRODUCT (drop-in kaldi_fbank_s on the GPU) against the third-party Kaldi fbank as the float64-grade truth of the suite's
    rule (tests/compare.py): within the tolerance of the oracle, or at the reference arithmetic's own float32 noise level FROM THE
    THIRD-PARTY VALUES (rms ratio <= 1.25, no single value beyond 2.5 x the CPU arithmetic's worst, the per-kind cap)"""
    from compare import check_parity
    from speech_amd import api
    x = signals.utterance(int(c["sr"] * 3.0), 21, sr=c["sr"], dtype=np.float32)
    o = dict(c)
    got = api.kaldi_fbank(x, sr=c["sr"], frame_length=c["fl"], frame_shift=c["fs"], B=c["B"], lof=c["lof"], hif=c["hif"])
    rc, ora = O.run("fbank", x, single=True, exact=True, **o)
    rc2, ora_fast = O.run("fbank", x, single=True, exact=False, **o)
    assert rc == 0 and rc2 == 0
    B = c["B"]
    K, _ = _kaldi_logspec(x.astype(np.float64), c["sr"], c["fl"], c["fs"], mel=_kaldi_mel(c["sr"], 1 << (int(round(c["sr"] * c["fl"] / 1000.0)) - 1).bit_length(), B, c["lof"], c["hif"]))
    K[:, B - 1] = K[:, B - 2]                                               # the reference's last-bin rule on Kaldi's values
    ok, e_go, e_gt, e_ot = check_parity("fbank", o, got, ora, truth=K, oracle_fast=ora_fast, label="third-party fbank " + IDS[CASES.index(c)])
    assert ok, (e_go, e_gt, e_ot)
    assert np.array_equal(got[:, B - 1], got[:, B - 2])
