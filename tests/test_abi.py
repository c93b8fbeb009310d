"""CPU tests of the drop-in boundary: libspeech.so loads, exports every symbol include/*.h declares, and reproduces
the reference's error convention (message on stderr, return 1; W==0 -> return 0) without touching a device.
No compute is called here (there is no GPU in this container and the library has no CPU fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from speech_amd import _lib, api

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(header):
    txt = open(os.path.join(ROOT, "include", header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return re.findall(r"^\s*(?:int|void|const char \*)\s*\*?\s*(\w+)\s*\(", txt, flags=re.M)


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    legacy = _declared("codee_speech.h")
    ext = _declared("speech_hip.h")
    assert sorted(legacy) == sorted(_lib.LEGACY_SYMBOLS)          # the reference's ten symbols: c/codee_speech.h:13-25
    assert sorted(ext) == sorted(_lib.EXT_SYMBOLS)
    for name in legacy + ext:
        assert hasattr(lib, name), name


def test_symbols_are_unmangled_c_symbols():
    import subprocess
    out = subprocess.check_output(["nm", "-D", "--defined-only", _lib.LIB_PATH]).decode()
    syms = {l.split()[-1] for l in out.splitlines() if " T " in l}
    for name in _lib.LEGACY_SYMBOLS + _lib.EXT_SYMBOLS:
        assert name in syms, name


def test_library_does_not_link_the_oracle():
    import subprocess
    out = subprocess.check_output(["nm", "-D", _lib.LIB_PATH]).decode()
    assert "oracle_" not in out
    ldd = subprocess.check_output(["ldd", _lib.LIB_PATH]).decode()
    assert "liboracle" not in ldd and "fftw" not in ldd


X = np.zeros(2000, dtype=np.float32)


@pytest.mark.parametrize("kw,msg", [
    (dict(sr=0.0), "sr must be positive"),
    (dict(frame_length=0.0), "frame_length must be positive"),
    (dict(frame_shift=0.0), "frame_shift must be positive"),
    (dict(dither=-1.0), "dither must be nonnegative"),
    (dict(preemph=-0.5), "preemph coeff must be nonnegative"),
    (dict(lof=-1.0), "lof (low-freq cutoff) must be nonnegative"),
    (dict(lof=50.0, hif=50.0), "hif (high-freq cutoff) must be > lof"),
    (dict(B=0), "B (num mel bins) must be positive"),
    (dict(C=0), "C (num cepstral coeffs) must be positive"),
    (dict(B=10, C=11), "C (num cepstral coeffs) must be <= B (num mel bins)"),
    (dict(Q=-1.0), "Q (lifter coeff) must be nonnegative"),
    (dict(frame_length=0.01), "frame_length must be greater than 1 sample"),
    (dict(frame_shift=0.01), "frame_shift must be greater than 1 sample"),
])
def test_argument_errors_match_reference_convention(kw, msg, capfd):
    """c/kaldi_mfcc.c:54-78: 'error in kaldi_mfcc_s: ...' on stderr and return 1 (validated before any device work)"""
    with pytest.raises(api.SpeechError):
        api.kaldi_mfcc(X, **kw)
    err = capfd.readouterr().err
    assert "error in kaldi_mfcc_s: " + msg in err


def test_double_entry_point_names_itself(capfd):
    with pytest.raises(api.SpeechError):
        api.kaldi_fbank(X.astype(np.float64), B=0)
    assert "error in kaldi_fbank_d: B (num mel bins) must be positive" in capfd.readouterr().err


def test_snip_edges_with_short_signal_is_an_error(capfd):
    lib = _lib.load()
    x = np.zeros(300, dtype=np.float32)
    y = np.zeros(13, dtype=np.float32)
    pf = C.POINTER(C.c_float)
    rc = lib.kaldi_mfcc_s(y.ctypes.data_as(pf), x.ctypes.data_as(pf), 300, 16000.0, 25.0, 10.0, 1, 0.0, 1, 1, 0.97, b"povey",
                          20.0, 8000.0, 23, 13, 22.0, 0, 0)
    assert rc == 1
    assert "frame length must be < signal length if snip_edges" in capfd.readouterr().err


def test_n_zero_is_an_error(capfd):
    lib = _lib.load()
    pf = C.POINTER(C.c_float)
    y = np.zeros(4, dtype=np.float32)
    rc = lib.kaldi_spectrogram_s(y.ctypes.data_as(pf), y.ctypes.data_as(pf), 0, 16000.0, 25.0, 10.0, 0, 0.0, 1, 1, 0.97, b"povey", 0)
    assert rc == 1
    assert "N (nsamps in signal) must be positive" in capfd.readouterr().err


def test_w_zero_returns_zero_and_leaves_y_untouched():
    """c/kaldi_mfcc.c:78: W==0 -> return 0 before any table is built (even the window name is not checked yet)"""
    lib = _lib.load()
    pf = C.POINTER(C.c_float)
    x = np.ones(50, dtype=np.float32)
    y = np.full(13, 7.0, dtype=np.float32)
    rc = lib.kaldi_mfcc_s(y.ctypes.data_as(pf), x.ctypes.data_as(pf), 50, 16000.0, 25.0, 10.0, 0, 0.0, 1, 1, 0.97, b"not-a-window",
                          20.0, 8000.0, 23, 13, 22.0, 0, 0)
    assert rc == 0
    assert (y == 7.0).all()


def test_frame_sizes_formula():
    assert api.frame_sizes(160000) == (998, 400, 160)
    assert api.frame_sizes(160000, snip_edges=0) == (1000, 400, 160)
    assert api.frame_sizes(22050, sr=22050.0)[1:] == (551, 220)     # 22050*25/1000 truncates in float32
    assert api.frame_sizes(22050, sr=22050.0, dtype=np.float64)[1:] == (551, 220)


def test_opts_struct_layout_matches_header():
    """speech_opts_default fills every field the ctypes mirror declares (guards against struct drift)."""
    lib = _lib.load()
    o = _lib.SpeechOpts()
    lib.speech_opts_default(C.byref(o), 2)
    assert (o.kind, o.dtype, o.sr, o.frame_length, o.frame_shift) == (2, 0, 16000.0, 25.0, 10.0)
    assert (o.snip_edges, o.dither, o.dc0, o.raw_energy, o.preemph) == (0, 0.0, 0, 0, 0.97)
    assert o.win_type == b"povey"
    assert (o.amp, o.lof, o.hif, o.B, o.lg, o.use_energy, o.C, o.Q) == (0, 20.0, 8000.0, 23, 1, 0, 13, 22.0)
    assert abs(o.compress - 1.0 / 3.0) < 1e-15 and o.cep_scale == 1.0 and o.lpc_order == 0 and o.mn0 == 0
    assert o.dither_seed == 0x5EED2026 and o.force_generic == 0


def test_plan_create_without_device_fails_loudly(have_gpu, capfd):
    if have_gpu:
        pytest.skip("a device is present")
    with pytest.raises(api.SpeechError) as e:
        api.Plan("mfcc")
    assert "no HIP device" in str(e.value)


def test_release_library_has_no_diagnostic_switches():
    """the SPEECH_ABLATE timing switches (kernels that skip loads, stores or whole phases) exist only in libspeech_diag.so
    (make -C speech_amd/csrc diag, -DSPEECH_DIAG): the shipped library neither names the variable nor reads the environment
    on its launch path.  The one getenv left is host_pipeline.hip's SPEECH_HOST_THREADS (packing threads, read once)."""
    import subprocess
    blob = open(_lib.LIB_PATH, "rb").read()
    assert b"SPEECH_ABLATE" not in blob
    names = [m.group(0) for m in re.finditer(rb"SPEECH_[A-Z_0-9]+", blob)]
    assert set(names) <= {b"SPEECH_HOST_THREADS", b"SPEECH_F64"}, set(names)      # SPEECH_F64: the enum's name in HIP_TRY messages
    # the launch path's translation unit (fast_kernels.o) makes no getenv call at all
    obj = os.path.join(ROOT, "speech_amd", "csrc", "fast_kernels.o")
    if os.path.exists(obj):
        und = subprocess.check_output(["nm", "-u", obj]).decode()
        assert "getenv" not in und


def test_c_sharder_equals_the_python_statement():
    """speech_shard_utterances (the sharder speech_run_host_multi uses) against api.shard_utterances on ragged lists with
    ties, single utterances and more shards than utterances"""
    rng = np.random.default_rng(11)
    for nutt, world in [(1, 1), (1, 4), (7, 3), (100, 8), (1000, 8), (257, 2), (64, 64), (5, 7)]:
        n = rng.integers(400, 480001, size=nutt)
        if nutt > 10:
            n[::5] = n[0]                                            # ties
        got = api.shard_assignment(n, world)
        want = np.full(nutt, -1)
        for r, idx in enumerate(api.shard_utterances(n, world)):
            want[idx] = r
        assert np.array_equal(got, want), (nutt, world)
    with pytest.raises(api.SpeechError):
        api.shard_assignment([1, 2, 3], 0)


def test_opts_frames_validates_and_sizes_without_a_device(capfd):
    """speech_opts_frames: W and D in the library's own arithmetic (c/kaldi_mfcc.c:68-78), and the reference's messages"""
    assert api.opts_frames("mfcc", 160000) == (998, 13)
    assert api.opts_frames("mfcc", 160000, snip=0) == (1000, 13)
    assert api.opts_frames("spectrogram", 22050, sr=22050.0) == (1 + (22050 - 551) // 220, 513)
    assert api.opts_frames("fbank", 16000, B=80, use_energy=1, dtype="f64") == (98, 81)
    assert api.opts_frames("plp", 100, snip=0) == (1, 13)
    for N in (400, 401, 559, 560, 561, 16000, 479999):
        for snip in (0, 1):
            assert api.opts_frames("mfcc", N, snip=snip)[0] == api.frame_sizes(N, snip_edges=snip)[0]
    with pytest.raises(api.SpeechError):
        api.opts_frames("mfcc", 300)
    assert "error in kaldi_mfcc_s: frame length must be < signal length if snip_edges" in capfd.readouterr().err
    with pytest.raises(api.SpeechError):
        api.opts_frames("fbank", 3000, win="kaiser", dtype="f64")
    assert "error in kaldi_fbank_d: window type must be" in capfd.readouterr().err
    with pytest.raises(api.SpeechError):
        api.opts_frames("mfcc", 0, snip=0)
    assert "N (nsamps in signal) must be positive" in capfd.readouterr().err


def test_run_host_multi_rejects_bad_arguments_before_touching_a_device(capfd):
    x = [np.zeros(2000, dtype=np.float32)]
    with pytest.raises(api.SpeechError):
        api.run_host_multi("mfcc", x, [])
    assert "at least one device" in capfd.readouterr().err
    with pytest.raises(api.SpeechError):
        api.run_host_multi("mfcc", x, [0], B=0)
    assert "error in kaldi_mfcc_s: B (num mel bins) must be positive" in capfd.readouterr().err
    assert api.run_host_multi("mfcc", [], [0]) == []
