"""CPU tests of the oracle (no GPU): C restatement vs the independent numpy float64 statement, the committed
golden fixtures, the reference's documented quirks, frame-count formulas and error returns."""
import numpy as np
import pytest

import np_truth as T
import oracle_lib as O
import signals
from cases import GRID, case_id
from compare import max_err

TRUTH = {"spectrogram": T.spectrogram, "fbank": T.fbank, "mfcc": T.mfcc, "plp": T.plp}


def _truth(kind, x, single, o):
    o = dict(o)
    mn0 = o.pop("mn0", 0)
    y = TRUTH[kind](np.asarray(x, dtype=np.float64), single=single, **o)
    return T.mean_norm(y) if mn0 else y


@pytest.mark.parametrize("case", GRID, ids=case_id)
def test_oracle_double_matches_numpy_truth(case):
    """The *_d restatement and the numpy statement are both float64: they must agree to rounding."""
    kind, N, seed, o = case
    x = signals.utterance(N, seed, sr=o.get("sr", 16000.0), dtype=np.float64)
    rc, y = O.run(kind, x, single=False, **o)
    assert rc == 0
    yt = _truth(kind, x, False, o)
    assert y.shape == yt.shape
    # near the FLT_EPSILON floor a log amplifies rounding; int16-scale audio keeps everything far above it
    np.testing.assert_allclose(y, yt, rtol=0, atol=5e-8 * max(1.0, np.abs(yt).max()))


@pytest.mark.parametrize("case", GRID, ids=case_id)
def test_oracle_single_close_to_truth(case):
    """The *_s restatement carries the reference's own float32 rounding noise (~1e-4)."""
    kind, N, seed, o = case
    x = signals.utterance(N, seed, sr=o.get("sr", 16000.0), dtype=np.float32)
    rc, y = O.run(kind, x, single=True, **o)
    assert rc == 0
    yt = _truth(kind, x, True, o)
    assert y.shape == yt.shape
    # Levinson-Durbin on weakly compressed mel energies is ill-conditioned in float32
    tol = 0.2 if (kind == "plp" and o.get("compress", 1.0 / 3.0) >= 0.5) else 2e-3
    assert max_err(kind, o, y, yt) < tol


def test_fast_fft_mode_close_to_exact():
    x = signals.utterance(8000, 9)
    for kind in TRUTH:
        _, a = O.run(kind, x, exact=True)
        _, b = O.run(kind, x, exact=False)
        # float32 FFT rounding alone moves cepstra by ~1.5e-4 on this signal: the reference's own float noise
        assert max_err(kind, {}, b, a) < 4e-4


def test_golden_fixtures():
    import os
    import importlib.util
    here = os.path.dirname(os.path.abspath(__file__))
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(here, "golden", "make_golden.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    gold = np.load(os.path.join(here, "golden", "oracle_golden.npz"))
    for name, kind, n, seed, single, opts in mg.CASES:
        x = signals.utterance(n, seed, sr=opts.get("sr", 16000.0), dtype=np.float32 if single else np.float64)
        rc, y = O.run(kind, x, single=single, exact=True, **opts)
        assert rc == 0
        # bit-exact on the machine that generated them; libm differences across hosts stay within an ulp or two of log/pow
        np.testing.assert_allclose(y, gold[name], rtol=0, atol=1e-5 if single else 1e-12, err_msg=name)


def test_quirks_duplicate_last_element():
    """c/kaldi_spectrogram.c:244-248, c/kaldi_fbank.c:349-350, c/kaldi_mfcc.c:329-333, c/kaldi_plp.c:399-403"""
    x = signals.utterance(3000, 5)
    for kind in ("spectrogram", "fbank", "mfcc", "plp"):
        _, y = O.run(kind, x)
        assert np.array_equal(y[:, -1], y[:, -2]), kind


def test_spectrogram_bin0_is_log_energy():
    """c/kaldi_spectrogram.c:236: Y[0] = rawe (pre or post energy), then log"""
    x = signals.utterance(2000, 6)
    _, y1 = O.run("spectrogram", x, raw_energy=1)
    _, y0 = O.run("spectrogram", x, raw_energy=0)
    fr, e_pre, e_post, _ = T.front_end(x.astype(np.float64), 16000.0, 25.0, 10.0, 1, 1, 0.97, "povey")
    np.testing.assert_allclose(y1[:, 0], np.log(e_pre), atol=1e-4)
    np.testing.assert_allclose(y0[:, 0], np.log(e_post), atol=1e-4)


def test_frame_counts():
    """W = 1+(N-L)/stp (snip_edges) or (N+stp/2)/stp; L,stp truncated in the working precision (c/kaldi_mfcc.c:68-78)"""
    assert O.frame_count(160000) == (998, 400, 160)
    assert O.frame_count(160000, snip=0) == (1000, 400, 160)
    assert O.frame_count(16000) == (98, 400, 160)
    assert O.frame_count(480000) == (2998, 400, 160)
    assert O.frame_count(400) == (1, 400, 160)
    assert O.frame_count(399) is None                 # snip_edges && L>N is an error
    assert O.frame_count(79, snip=0) == (0, 400, 160)  # W==0 -> return 0
    assert O.frame_count(80, snip=0) == (1, 400, 160)
    W, L, stp = O.frame_count(22050, sr=22050.0)
    assert (L, stp) == (551, 220)


def test_w0_returns_zero_and_leaves_y():
    x = signals.utterance(50, 1)
    rc, y = O.run("mfcc", x, snip=0)
    assert rc == 0 and y is None


@pytest.mark.parametrize("bad", [
    dict(sr=0.0), dict(fl=0.0), dict(fs=0.0), dict(dither=-1.0), dict(preemph=-0.1), dict(lof=-1.0),
    dict(hif=10.0, lof=20.0), dict(B=0), dict(C=0), dict(C=30, B=23), dict(Q=-1.0), dict(win="kaiser"),
    dict(fl=0.01), dict(fs=0.01),
])
def test_error_returns(bad, capfd):
    """c/kaldi_mfcc.c:54-78,131-134: message on stderr, return 1"""
    x = signals.utterance(2000, 1)
    rc, y = O.run("mfcc", x, **bad)
    assert rc == 1 and y is None
    assert "error in kaldi_mfcc_s" in capfd.readouterr().err


def test_snip_edges_short_signal_is_error(capfd):
    x = signals.utterance(300, 1)
    rc, _ = O.run("fbank", x, snip=1)
    assert rc == 1
    assert "frame length must be < signal length if snip_edges" in capfd.readouterr().err


def test_reflection_indexing():
    """c/kaldi_mfcc.c:228: n<0 -> -n-1, n>=N -> 2N-1-n, iterated"""
    x = np.arange(10, dtype=np.float64)
    fr = T.frames(x, 8, 3, 3, 0)
    # frame 0 starts at stp/2 - L/2 = 1 - 4 = -3: samples -3,-2,-1 reflect to 2,1,0
    assert list(fr[0]) == [2, 1, 0, 0, 1, 2, 3, 4]
    x3 = np.arange(3, dtype=np.float64)
    fr = T.frames(x3, 8, 3, 1, 0)
    assert list(fr[0]) == [2, 1, 0, 0, 1, 2, 2, 1]


def test_batch_driver_matches_single_calls():
    import ctypes as C
    lib = O.lib()
    lib.oracle_set_fft_exact(0)
    ns = [4000, 2500, 16000]
    xs = [signals.utterance(n, 50 + i) for i, n in enumerate(ns)]
    X = np.concatenate(xs)
    xo = np.array([0, 4000, 6500], dtype=np.uint64)
    Ws = [O.frame_count(n)[0] for n in ns]
    yo = np.array([0, Ws[0] * 13, (Ws[0] + Ws[1]) * 13], dtype=np.uint64)
    Y = np.zeros(sum(Ws) * 13, dtype=np.float32)
    nn = np.array(ns, dtype=np.uint64)
    pz, pf = C.POINTER(C.c_size_t), C.POINTER(C.c_float)
    used = lib.oracle_batch_s(2, 3, X.ctypes.data_as(pf), xo.ctypes.data_as(pz), nn.ctypes.data_as(pz), Y.ctypes.data_as(pf),
                              yo.ctypes.data_as(pz), 16000.0, 25.0, 10.0, 1, 1, 1, 0.97, b"povey", 0, 20.0, 8000.0, 23, 1, 13, 22.0,
                              1.0 / 3.0, 1.0, 0, 2)
    assert used >= 1
    for i, x in enumerate(xs):
        _, y = O.run("mfcc", x, exact=False)
        got = Y[int(yo[i]):int(yo[i]) + y.size].reshape(y.shape)
        assert np.array_equal(got, y)
    lib.oracle_set_fft_exact(1)


def test_oracle_is_clean_under_address_and_ub_sanitizers():
    """oracle/san_driver.c: every extractor over the edge cases where the reference has out-of-bounds hazards (tiny frames,
    B = 1, C = 1, one-sample signals, hif above Nyquist), outputs sized exactly W x D, built with -fsanitize=address,undefined"""
    import os
    import subprocess
    odir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle")
    subprocess.check_call(["make", "-C", odir, "san"], stdout=subprocess.DEVNULL)
    env = dict(os.environ, ASAN_OPTIONS="verify_asan_link_order=0:detect_leaks=1", UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([os.path.join(odir, "_san", "oracle_san")], capture_output=True, env=env, timeout=600)
    err = r.stderr.decode()
    assert r.returncode == 0, err[-3000:]
    assert "AddressSanitizer" not in err and "runtime error" not in err and "LeakSanitizer" not in err
    assert b"0 failure(s)" in r.stdout
