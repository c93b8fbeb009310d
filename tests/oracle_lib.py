"""ctypes loader for the CPU oracle (oracle/liboracle.so).  Test infrastructure only: nothing in
speech_amd/ imports this.  If the library is missing it is built with oracle/Makefile (gcc)."""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_DIR = os.path.join(os.path.dirname(_HERE), "oracle")

_f, _d, _i, _z, _s = C.c_float, C.c_double, C.c_int, C.c_size_t, C.c_char_p
_pf, _pd, _pz = C.POINTER(C.c_float), C.POINTER(C.c_double), C.POINTER(C.c_size_t)


def _load(name="liboracle.so"):
    path = os.path.join(ORACLE_DIR, name)
    if not os.path.exists(path):
        subprocess.check_call(["make", "-C", ORACLE_DIR, name])
    lib = C.CDLL(path)
    for sfx, r, p in (("s", _f, _pf), ("d", _d, _pd)):
        front = [p, p, _z, r, r, r, _i, r, _i, _i, r, _s]
        getattr(lib, "oracle_kaldi_spectrogram_" + sfx).argtypes = front + [_i]
        getattr(lib, "oracle_kaldi_fbank_" + sfx).argtypes = front + [_i, r, r, _z, _i, _i, _i]
        getattr(lib, "oracle_kaldi_mfcc_" + sfx).argtypes = front + [r, r, _z, _z, r, _i, _i]
        getattr(lib, "oracle_kaldi_plp_" + sfx).argtypes = front + [r, r, _z, r, _z, r, r, _i, _i, _z]
        for k in ("spectrogram", "fbank", "mfcc", "plp"):
            getattr(lib, "oracle_kaldi_%s_%s" % (k, sfx)).restype = _i
        getattr(lib, "oracle_frame_count_" + sfx).argtypes = [_z, r, r, r, _i, _pz, _pz, _pz]
        getattr(lib, "oracle_frame_count_" + sfx).restype = _i
    lib.oracle_set_fft_exact.argtypes = [_i]
    lib.oracle_batch_s.argtypes = [_i, _z, _pf, _pz, _pz, _pf, _pz, _f, _f, _f, _i, _i, _i, _f, _s,
                                   _i, _f, _f, _z, _i, _z, _f, _f, _f, _i, _i]
    lib.oracle_batch_s.restype = _i
    return lib


_LIBS = {}


def lib(name="liboracle.so"):
    if name not in _LIBS:
        _LIBS[name] = _load(name)
    return _LIBS[name]


def frame_count(N, sr=16000.0, fl=25.0, fs=10.0, snip=1, single=True):
    W, L, stp = C.c_size_t(0), C.c_size_t(0), C.c_size_t(0)
    fn = lib().oracle_frame_count_s if single else lib().oracle_frame_count_d
    rc = fn(N, sr, fl, fs, snip, C.byref(W), C.byref(L), C.byref(stp))
    if rc:
        return None
    return W.value, L.value, stp.value


def _nfft(L):
    n = 1
    while n < L:
        n *= 2
    return n


DEFAULTS = dict(sr=16000.0, fl=25.0, fs=10.0, snip=1, dither=0.0, dc0=1, raw_energy=1, preemph=0.97, win="povey",
                amp=0, lof=20.0, hif=8000.0, B=23, lg=1, C=13, Q=22.0, compress=1.0 / 3.0, cep_scale=1.0,
                use_energy=0, mn0=0, order=0)


def feat_dim(kind, L, o):
    if kind == "spectrogram":
        return _nfft(L) // 2 + 1
    if kind == "fbank":
        return o["B"] + (1 if o["use_energy"] else 0)
    return o["C"]


def run(kind, x, single=True, exact=True, **kw):
    """Run the oracle on one utterance.  Returns (rc, Y or None)."""
    o = dict(DEFAULTS)
    o.update(kw)
    dt = np.float32 if single else np.float64
    x = np.ascontiguousarray(x, dtype=dt)
    l = lib()
    l.oracle_set_fft_exact(1 if exact else 0)
    fc = frame_count(len(x), o["sr"], o["fl"], o["fs"], o["snip"], single)
    W, L = (fc[0], fc[1]) if fc else (0, 1)
    D = feat_dim(kind, max(L, 1), o)
    Y = np.full((max(W, 1), D), np.nan, dtype=dt)
    p = _pf if single else _pd
    sfx = "s" if single else "d"
    front = [Y.ctypes.data_as(p), x.ctypes.data_as(p), len(x), o["sr"], o["fl"], o["fs"], o["snip"], o["dither"],
             o["dc0"], o["raw_energy"], o["preemph"], o["win"].encode()]
    fn = getattr(l, "oracle_kaldi_%s_%s" % (kind, sfx))
    if kind == "spectrogram":
        rc = fn(*front, o["mn0"])
    elif kind == "fbank":
        rc = fn(*front, o["amp"], o["lof"], o["hif"], o["B"], o["lg"], o["use_energy"], o["mn0"])
    elif kind == "mfcc":
        rc = fn(*front, o["lof"], o["hif"], o["B"], o["C"], o["Q"], o["use_energy"], o["mn0"])
    elif kind == "plp":
        rc = fn(*front, o["lof"], o["hif"], o["B"], o["compress"], o["C"], o["Q"], o["cep_scale"], o["use_energy"], o["mn0"], o["order"])
    else:
        raise ValueError(kind)
    if rc or W == 0:
        return rc, None
    return rc, Y[:W]
