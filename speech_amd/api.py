"""Host-side mirror of the reference interface, over the C ABI.

Two levels, both thin:
  * kaldi_spectrogram / kaldi_fbank / kaldi_mfcc / kaldi_plp -- the reference's per-utterance functions
    (c/codee_speech.h:13-25) with the same argument names and meaning; numpy in, numpy out, host pointers
    straight into the drop-in symbols.  A non-zero return raises SpeechError (the C side has already printed
    the reference-style message to stderr).
  * Plan / Batch -- the device-resident ragged-batch ABI of include/speech_hip.h, taking torch CUDA (ROCm)
    tensors; torch is used only for device memory and streams.
"""
import ctypes as C
import numpy as np
from . import _lib


class SpeechError(RuntimeError):
    pass


def _nfft(L):
    n = 1
    while n < L:
        n *= 2
    return n


def frame_sizes(N, sr=16000.0, frame_length=25.0, frame_shift=10.0, snip_edges=1, dtype=np.float32):
    """(W, L, stp) as the reference derives them in the working precision (c/kaldi_mfcc.c:68-78)."""
    t = np.float32 if np.dtype(dtype) == np.float32 else np.float64
    L = int(t(sr) * t(frame_length) / t(1000.0))
    stp = int(t(sr) * t(frame_shift) / t(1000.0))
    if L < 1 or stp < 1 or N < 1 or (snip_edges and L > N):
        return 0, L, stp
    W = 1 + (N - L) // stp if snip_edges else (N + stp // 2) // stp
    return W, L, stp


def _legacy(kind, X, D_of, tail, sr, frame_length, frame_shift, snip_edges, dither, dc0, raw_energy, preemph, win_type, mn0):
    lib = _lib.load()
    X = np.ascontiguousarray(X)
    if X.dtype not in (np.float32, np.float64):
        X = X.astype(np.float32)
    X = X.reshape(-1)
    single = X.dtype == np.float32
    ct = C.c_float if single else C.c_double
    W, L, stp = frame_sizes(X.size, sr, frame_length, frame_shift, snip_edges, X.dtype)
    D = D_of(max(L, 1))
    Y = np.zeros((max(W, 1), D), dtype=X.dtype)
    fn = getattr(lib, "kaldi_%s_%s" % (kind, "s" if single else "d"))
    rc = fn(Y.ctypes.data_as(C.POINTER(ct)), X.ctypes.data_as(C.POINTER(ct)), X.size, sr, frame_length, frame_shift,
            int(snip_edges), dither, int(dc0), int(raw_energy), preemph, win_type.encode(), *tail, int(mn0))
    if rc != 0:
        raise SpeechError(_lib.last_error() or "kaldi_%s failed" % kind)
    return Y[:W]


def kaldi_spectrogram(X, sr=16000.0, frame_length=25.0, frame_shift=10.0, snip_edges=1, dither=0.0, dc0=1, raw_energy=1,
                      preemph=0.97, win_type="povey", mn0=0):
    """reference: c/kaldi_spectrogram.c:46 (float32 input -> *_s, float64 -> *_d)"""
    return _legacy("spectrogram", X, lambda L: _nfft(L) // 2 + 1, (), sr, frame_length, frame_shift, snip_edges, dither,
                   dc0, raw_energy, preemph, win_type, mn0)


def kaldi_fbank(X, sr=16000.0, frame_length=25.0, frame_shift=10.0, snip_edges=1, dither=0.0, dc0=1, raw_energy=1,
                preemph=0.97, win_type="povey", amp=0, lof=20.0, hif=8000.0, B=23, lg=1, use_energy=0, mn0=0):
    """reference: c/kaldi_fbank.c:59"""
    return _legacy("fbank", X, lambda L: B + (1 if use_energy else 0), (int(amp), lof, hif, B, int(lg), int(use_energy)),
                   sr, frame_length, frame_shift, snip_edges, dither, dc0, raw_energy, preemph, win_type, mn0)


def kaldi_mfcc(X, sr=16000.0, frame_length=25.0, frame_shift=10.0, snip_edges=1, dither=0.0, dc0=1, raw_energy=1,
               preemph=0.97, win_type="povey", lof=20.0, hif=8000.0, B=23, C=13, Q=22.0, use_energy=0, mn0=0):
    """reference: c/kaldi_mfcc.c:52"""
    return _legacy("mfcc", X, lambda L: C, (lof, hif, B, C, Q, int(use_energy)),
                   sr, frame_length, frame_shift, snip_edges, dither, dc0, raw_energy, preemph, win_type, mn0)


def kaldi_plp(X, sr=16000.0, frame_length=25.0, frame_shift=10.0, snip_edges=1, dither=0.0, dc0=1, raw_energy=1,
              preemph=0.97, win_type="povey", lof=20.0, hif=8000.0, B=23, compress=1.0 / 3.0, C=13, Q=22.0, cep_scale=1.0,
              use_energy=0, mn0=0):
    """reference: c/kaldi_plp.c:53 (signature c/codee_speech.h:21)"""
    return _legacy("plp", X, lambda L: C, (lof, hif, B, compress, C, Q, cep_scale, int(use_energy)),
                   sr, frame_length, frame_shift, snip_edges, dither, dc0, raw_energy, preemph, win_type, mn0)


_OPT_ALIASES = {"fl": "frame_length", "fs": "frame_shift", "snip": "snip_edges", "win": "win_type", "order": "lpc_order"}


def make_opts(kind, dtype="f32", **kw):
    lib = _lib.load()
    o = _lib.SpeechOpts()
    lib.speech_opts_default(C.byref(o), _lib.KINDS[kind])
    o.dtype = _lib.F64 if dtype in ("f64", np.float64, "double") else _lib.F32
    # test/bench convention: Kaldi-style front end unless overridden (the CLI defaults leave dc0/raw_energy/snip_edges off)
    o.snip_edges, o.dc0, o.raw_energy = 1, 1, 1
    for k, v in kw.items():
        k = _OPT_ALIASES.get(k, k)
        if k == "win_type":
            o.win_type = v.encode()
        elif not hasattr(o, k):
            raise KeyError(k)
        else:
            setattr(o, k, v)
    return o


def extract_pcm16(kind, X16, dtype="f32", **kw):
    """One utterance of 16-bit PCM (numpy int16, host) -> features, as the drop-in call on the converted audio returns
    them; half the bytes cross PCIe (speech_extract_pcm16, include/speech_hip.h).  Options as make_opts."""
    lib = _lib.load()
    o = make_opts(kind, dtype, **kw)
    X16 = np.ascontiguousarray(X16)
    if X16.dtype != np.int16:
        raise TypeError("X16 must be int16")
    X16 = X16.reshape(-1)
    t = np.float64 if o.dtype == _lib.F64 else np.float32
    W, L, _ = frame_sizes(X16.size, o.sr, o.frame_length, o.frame_shift, o.snip_edges, t)
    D = {"spectrogram": _nfft(max(L, 1)) // 2 + 1, "fbank": o.B + (1 if o.use_energy else 0)}.get(kind, o.C)
    Y = np.zeros((max(W, 1), D), dtype=t)
    if lib.speech_extract_pcm16(C.byref(o), X16.ctypes.data_as(C.POINTER(C.c_short)), X16.size, Y.ctypes.data_as(C.c_void_p)) != 0:
        raise SpeechError(_lib.last_error() or "speech_extract_pcm16 failed")
    return Y[:W]


def opts_frames(kind, N, dtype="f32", **kw):
    """(W, D) of an utterance of N samples under these options: speech_opts_frames -- validation and sizes in the library's
    own arithmetic, no plan, no device.  Raises SpeechError where the reference's prologue would return 1."""
    lib = _lib.load()
    o = make_opts(kind, dtype, **kw)
    W, D = C.c_size_t(), C.c_size_t()
    if lib.speech_opts_frames(C.byref(o), int(N), C.byref(W), C.byref(D)) != 0:
        raise SpeechError(_lib.last_error())
    return W.value, D.value


def shard_assignment(n_samples, nshards):
    """speech_shard_utterances: the library's own (C) statement of shard_utterances below, as one shard number per
    utterance.  Host arithmetic only."""
    lib = _lib.load()
    n = np.ascontiguousarray(n_samples, dtype=np.uint64)
    out = np.zeros(max(1, n.size), dtype=np.int32)
    if lib.speech_shard_utterances(n.size, n.ctypes.data_as(C.POINTER(C.c_size_t)), int(nshards), out.ctypes.data_as(C.POINTER(C.c_int))) != 0:
        raise SpeechError(_lib.last_error())
    return out[:n.size]


def run_host_multi(kind, xs, devices, dtype="f32", chunk_bytes=0, ids=None, **kw):
    """speech_run_host_multi: a list of utterances in host memory (numpy arrays of `dtype`, or all int16) over the HIP
    devices `devices` from ONE call -- a host thread, a plan and a two-stream staging pipeline per device inside the library,
    LPT sharding, no collective.  Returns the list of W_u x D arrays; equal bit for bit to Plan.run_host on one device."""
    lib = _lib.load()
    o = make_opts(kind, dtype, **kw)
    t = np.float64 if o.dtype == _lib.F64 else np.float32
    n = len(xs)
    pcm16 = n > 0 and all(np.asarray(x).dtype == np.int16 for x in xs)
    xs = [np.ascontiguousarray(x, dtype=np.int16 if pcm16 else t).reshape(-1) for x in xs]
    ys = []
    W, D = C.c_size_t(), C.c_size_t()
    for x in xs:
        if lib.speech_opts_frames(C.byref(o), x.size, C.byref(W), C.byref(D)) != 0:
            raise SpeechError(_lib.last_error())
        ys.append(np.zeros((W.value, D.value), dtype=t))
    dev = (C.c_int * max(1, len(devices)))(*[int(d) for d in devices])
    Xp = (C.c_void_p * max(1, n))(*[x.ctypes.data for x in xs])
    Yp = (C.c_void_p * max(1, n))(*[y.ctypes.data if y.size else xs[0].ctypes.data for y in ys])
    Np = (C.c_size_t * max(1, n))(*[x.size for x in xs])
    idp = None
    if ids is not None:
        ida = np.ascontiguousarray(ids, dtype=np.uint32)
        if ida.shape != (n,):
            raise ValueError("ids must have one entry per utterance")
        idp = ida.ctypes.data_as(C.POINTER(C.c_uint))
    if lib.speech_run_host_multi(C.byref(o), len(devices), dev, n, Xp, Np, Yp, int(pcm16), int(chunk_bytes), idp) != 0:
        raise SpeechError(_lib.last_error())
    return ys


class Plan:
    """speech_plan: the per-call tables of c/kaldi_mfcc.c:105-210, built once on one device."""

    def __init__(self, kind, device=0, dtype="f32", **kw):
        self._lib = _lib.load()
        self.opts = make_opts(kind, dtype, **kw)
        self.kind = kind
        self.device = device
        h = C.c_void_p()
        if self._lib.speech_plan_create(C.byref(h), C.byref(self.opts), device) != 0:
            raise SpeechError(_lib.last_error())
        self._h = h
        L, stp, nfft, D, tuned = C.c_size_t(), C.c_size_t(), C.c_size_t(), C.c_size_t(), C.c_int()
        self._lib.speech_plan_sizes(h, C.byref(L), C.byref(stp), C.byref(nfft), C.byref(D), C.byref(tuned))
        self.L, self.stp, self.nfft, self.D, self.tuned = L.value, stp.value, nfft.value, D.value, bool(tuned.value)
        self.kernel = ("generic", "tuned", "quad_f64")[tuned.value]      # which frame kernel serves the plan (include/speech_hip.h)
        self.np_dtype = np.float64 if self.opts.dtype == _lib.F64 else np.float32

    def num_frames(self, N):
        W = C.c_size_t()
        if self._lib.speech_plan_num_frames(self._h, N, C.byref(W)) != 0:
            return None
        return W.value

    def run_host(self, xs, chunk_bytes=0, out=None, ids=None):
        """features of a list of utterances held in host memory (numpy arrays of the plan's dtype, or all int16 = 16-bit PCM),
        as a list of W_u x D numpy arrays: speech_plan_run_host, chunked and overlapped host -> device -> host.  ids: the
        utterances' numbers in the corpus, for the dither generator (default: their positions in xs)"""
        n = len(xs)
        pcm16 = n > 0 and all(np.asarray(x).dtype == np.int16 for x in xs)
        dt = np.int16 if pcm16 else self.np_dtype
        xs = [np.ascontiguousarray(x, dtype=dt).reshape(-1) for x in xs]
        ys = out if out is not None else []
        if out is None:
            for x in xs:
                W = self.num_frames(x.size)
                if W is None:
                    raise SpeechError("an utterance of %d samples is not valid for this plan" % x.size)
                ys.append(np.zeros((W, self.D), dtype=self.np_dtype))
        else:
            if len(ys) != n:
                raise ValueError("out must hold one array per utterance")
            for x, y in zip(xs, ys):
                W = self.num_frames(x.size)                   # None: the C side rejects the utterance with the reference's message
                if W is not None and (not isinstance(y, np.ndarray) or y.dtype != self.np_dtype or not y.flags.c_contiguous or y.size < W * self.D):
                    raise ValueError("out arrays must be C-contiguous %s arrays of at least W x D = %d x %d elements" % (np.dtype(self.np_dtype).name, W, self.D))
        Xp = (C.c_void_p * max(1, n))(*[x.ctypes.data for x in xs])
        Yp = (C.c_void_p * max(1, n))(*[y.ctypes.data if y.size else xs[0].ctypes.data for y in ys])
        Np = (C.c_size_t * max(1, n))(*[x.size for x in xs])
        idp = None
        if ids is not None:
            ida = np.ascontiguousarray(ids, dtype=np.uint32)
            if ida.shape != (n,):
                raise ValueError("ids must have one entry per utterance")
            idp = ida.ctypes.data_as(C.POINTER(C.c_uint))
        if self._lib.speech_plan_run_host(self._h, n, Xp, Np, Yp, int(pcm16), int(chunk_bytes), idp) != 0:
            raise SpeechError(_lib.last_error())
        return ys

    def close(self):
        if getattr(self, "_h", None):
            self._lib.speech_plan_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _dev_ptr(t, what, plan, numel, dtypes=None):
    """data_ptr() of a torch tensor after checking what the C ABI cannot: device, dtype, contiguity and size (a wrong
    tensor would otherwise be a GPU memory fault, not an error)."""
    name = str(t.dtype).replace("torch.", "")
    want = dtypes or ("float64" if plan.np_dtype == np.float64 else "float32",)
    if not t.is_cuda or (t.device.index or 0) != plan.device:
        raise ValueError("%s must be on cuda:%d (the plan's device), got %s" % (what, plan.device, t.device))
    if name not in want:
        raise ValueError("%s must have dtype %s, got %s" % (what, " or ".join(want), name))
    if not t.is_contiguous():
        raise ValueError("%s must be contiguous" % what)
    if t.numel() < numel:
        raise ValueError("%s holds %d elements, the batch needs %d" % (what, t.numel(), numel))
    return C.c_void_p(t.data_ptr())


class Batch:
    """speech_batch: launch geometry for a ragged list of utterances packed back to back."""

    def __init__(self, plan, n_samples, x_off=None, y_off=None):
        self._lib = _lib.load()
        self.plan = plan
        n = np.ascontiguousarray(n_samples, dtype=np.uint64)
        self.nutt = int(n.size)
        pz = C.POINTER(C.c_size_t)
        xo = np.ascontiguousarray(x_off, dtype=np.uint64) if x_off is not None else None
        yo = np.ascontiguousarray(y_off, dtype=np.uint64) if y_off is not None else None
        h = C.c_void_p()
        rc = self._lib.speech_batch_create(C.byref(h), plan._h, self.nutt, n.ctypes.data_as(pz),
                                           xo.ctypes.data_as(pz) if xo is not None else None,
                                           yo.ctypes.data_as(pz) if yo is not None else None)
        if rc != 0:
            raise SpeechError(_lib.last_error())
        self._h = h
        ts, to, tf = C.c_size_t(), C.c_size_t(), C.c_size_t()
        self._lib.speech_batch_totals(h, C.byref(ts), C.byref(to), C.byref(tf))
        self.total_samples, self.total_out, self.total_frames = ts.value, to.value, tf.value
        self.x_off = np.zeros(self.nutt, dtype=np.uint64)
        self.y_off = np.zeros(self.nutt, dtype=np.uint64)
        self.n_frames = np.zeros(self.nutt, dtype=np.uint64)
        if self.nutt:
            self._lib.speech_batch_offsets(h, self.x_off.ctypes.data_as(pz), self.y_off.ctypes.data_as(pz), self.n_frames.ctypes.data_as(pz))
        self.n_samples = n
        # one past the last element the kernels touch in X / Y (checked against the tensors handed to run)
        self._x_end = int((self.x_off + n).max()) if self.nutt else 0
        self._y_end = int((self.y_off + self.n_frames * np.uint64(plan.D)).max()) if self.nutt else 0

    def run(self, X, Y, stream=None, timed=False):
        """X, Y: torch tensors on the plan's device (1-D, plan dtype; X may also be int16 PCM).  stream: a raw hipStream_t value
        (e.g. torch.cuda.current_stream().cuda_stream) or None for the default stream."""
        sp = C.c_void_p(stream) if stream else None
        need_x = int(self._x_end) if self.nutt else 0
        need_y = int(self._y_end) if self.nutt else 0
        yp = _dev_ptr(Y, "Y", self.plan, need_y)
        if str(X.dtype).endswith("int16"):
            # 16-bit PCM on the device (speech_batch_run_pcm16): converted to the plan's dtype as it is loaded
            ms = C.c_float()
            rc = self._lib.speech_batch_run_pcm16(self._h, _dev_ptr(X, "X", self.plan, need_x, ("int16",)), yp, sp, C.byref(ms) if timed else None)
            if rc != 0:
                raise SpeechError(_lib.last_error())
            return ms.value if timed else None
        xp = _dev_ptr(X, "X", self.plan, need_x)
        if timed:
            ms = C.c_float()
            rc = self._lib.speech_batch_run_timed(self._h, xp, yp, sp, C.byref(ms))
            if rc != 0:
                raise SpeechError(_lib.last_error())
            return ms.value
        rc = self._lib.speech_batch_run(self._h, xp, yp, sp)
        if rc != 0:
            raise SpeechError(_lib.last_error())
        return None

    def synth_audio(self, X, seed=0x5EED2026, ids=None, stream=None):
        """fill X (plan dtype, or int16) with the synthetic corpus of SURVEY.md 8(d), generated on the device: utterance u of
        the batch is corpus utterance ids[u] (default u); the same samples as corpus.synth_utterance gives on the host"""
        pcm = str(X.dtype).endswith("int16")
        xp = _dev_ptr(X, "X", self.plan, int(self._x_end) if self.nutt else 0, ("int16",) if pcm else None)
        idp = None
        if ids is not None:
            ida = np.ascontiguousarray(ids, dtype=np.uint64)
            if ida.shape != (self.nutt,):
                raise ValueError("ids must have one entry per utterance")
            idp = ida.ctypes.data_as(C.POINTER(C.c_ulonglong))
        self._check(self._lib.speech_batch_synth_audio(self._h, xp, int(pcm), int(seed), idp, C.c_void_p(stream) if stream else None))

    # ---- corpus post-processing on the packed features (include/speech_hip.h) ----

    def _check(self, rc):
        if rc != 0:
            raise SpeechError(_lib.last_error())

    def stats(self, Y, stream=None):
        """per-feature (sum, sum of squares, frames) over the batch, float64 numpy arrays of D; waits for the stream"""
        D = self.plan.D
        s, q, n = np.zeros(D), np.zeros(D), C.c_ulonglong()
        pd = C.POINTER(C.c_double)
        self._check(self._lib.speech_batch_stats(self._h, _dev_ptr(Y, "Y", self.plan, self._y_end), s.ctypes.data_as(pd), q.ctypes.data_as(pd), C.byref(n),
                                                 C.c_void_p(stream) if stream else None))
        return s, q, int(n.value)

    def cmvn(self, Y, norm_vars=False, stream=None):
        """per-utterance mean (and variance) normalisation, in place"""
        self._check(self._lib.speech_batch_cmvn(self._h, _dev_ptr(Y, "Y", self.plan, self._y_end), int(bool(norm_vars)), C.c_void_p(stream) if stream else None))

    def cmvn_noalloc(self, Y, norm_vars=False, stream=None):
        """cmvn through speech_batch_cmvn_noalloc: the same result from a static 4 MiB device scratch, no allocation"""
        self._check(self._lib.speech_batch_cmvn_noalloc(self._h, _dev_ptr(Y, "Y", self.plan, self._y_end), int(bool(norm_vars)),
                                                        C.c_void_p(stream) if stream else None))

    def affine(self, Y, scale, offset, stream=None):
        """y = y * scale[d] + offset[d] in place (e.g. corpus-wide CMVN from merged stats)"""
        D = self.plan.D
        sc, of = np.ascontiguousarray(scale, dtype=np.float64), np.ascontiguousarray(offset, dtype=np.float64)
        if sc.shape != (D,) or of.shape != (D,):
            raise ValueError("scale and offset must have shape (%d,)" % D)
        pd = C.POINTER(C.c_double)
        self._check(self._lib.speech_batch_affine(self._h, _dev_ptr(Y, "Y", self.plan, self._y_end), sc.ctypes.data_as(pd), of.ctypes.data_as(pd),
                                                  C.c_void_p(stream) if stream else None))

    def add_deltas(self, Y, Z, order=2, window=2, stream=None):
        """Z (total_out * (order + 1) elements) = features with delta .. delta^order appended, Kaldi add-deltas"""
        self._check(self._lib.speech_batch_add_deltas(self._h, _dev_ptr(Y, "Y", self.plan, self._y_end), _dev_ptr(Z, "Z", self.plan, self._y_end * (int(order) + 1)), int(order), int(window),
                                                      C.c_void_p(stream) if stream else None))

    def splice(self, Y, Z, left=4, right=4, stream=None):
        """Z (total_out * (left + right + 1) elements) = frames spliced with their neighbours, Kaldi splice-feats"""
        self._check(self._lib.speech_batch_splice(self._h, _dev_ptr(Y, "Y", self.plan, self._y_end), _dev_ptr(Z, "Z", self.plan, self._y_end * (int(left) + int(right) + 1)), int(left), int(right),
                                                  C.c_void_p(stream) if stream else None))

    def set_ids(self, ids=None):
        """dither: utterance u draws its noise as utterance number ids[u] (None: its index in the batch)"""
        if ids is None:
            self._check(self._lib.speech_batch_set_ids(self._h, None))
            return
        a = np.ascontiguousarray(ids, dtype=np.uint32)
        if a.shape != (self.nutt,):
            raise ValueError("ids must have one entry per utterance")
        self._check(self._lib.speech_batch_set_ids(self._h, a.ctypes.data_as(C.POINTER(C.c_uint))))

    def subsample_rows(self, n, offset=0):
        """(rows per utterance, total rows) of splice_subsample's output for --n=n --offset=offset"""
        rows, tot = (C.c_size_t * max(1, self.nutt))(), C.c_size_t()
        self._check(self._lib.speech_batch_subsample_rows(self._h, int(n), int(offset), rows, C.byref(tot)))
        return np.array(rows[:self.nutt], dtype=np.int64), int(tot.value)

    def splice_subsample(self, Y, Z, left=0, right=0, n=3, offset=0, stream=None):
        """Z (subsample_rows(n, offset)[1] * D * (left + right + 1) elements, utterances back to back) = every n-th spliced
        frame from `offset`: Kaldi splice-feats | subsample-feats in one pass"""
        need_z = self.subsample_rows(n, offset)[1] * self.plan.D * (int(left) + int(right) + 1)
        self._check(self._lib.speech_batch_splice_subsample(self._h, _dev_ptr(Y, "Y", self.plan, self._y_end), _dev_ptr(Z, "Z", self.plan, need_z), int(left), int(right),
                                                            int(n), int(offset), C.c_void_p(stream) if stream else None))

    def close(self):
        if getattr(self, "_h", None):
            self._lib.speech_batch_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def merge_stats(s, q, n, group=None):
    """Corpus-wide statistics over ranks: the one exchange step of this path.  Adds the per-rank
    (sum, sumsq, frames) of Batch.stats with ONE all-reduce of 2 D + 1 float64 values over `group` (torch.distributed:
    RCCL between GPUs, gloo in the CPU tests); without an initialised process group it returns its inputs."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return np.asarray(s, dtype=np.float64), np.asarray(q, dtype=np.float64), int(n)
    D = len(s)
    buf = torch.from_numpy(np.concatenate([np.asarray(s, np.float64), np.asarray(q, np.float64), [float(n)]]))
    if dist.get_backend(group) == "nccl":
        buf = buf.cuda()
    dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
    out = buf.cpu().numpy()
    return out[:D].copy(), out[D:2 * D].copy(), int(round(out[2 * D]))


def cmvn_from_stats(s, q, n, norm_vars=True, var_floor=1.0e-20):
    """(scale, offset) of apply-cmvn from merged statistics: y = (x - mean) / std = x * scale + offset"""
    s, q = np.asarray(s, np.float64), np.asarray(q, np.float64)
    mean = s / float(n)
    scale = np.ones_like(mean)
    if norm_vars:
        scale = 1.0 / np.sqrt(np.maximum(q / float(n) - mean * mean, var_floor))
    return scale, -mean * scale


def extract_shard(plan, xs, rank, world_size, chunk_bytes=0):
    """This rank's part of a corpus held in host memory: (indices, features) of the utterances shard_utterances gives to
    `rank`, extracted with speech_plan_run_host under their corpus numbers -- so the union over the ranks equals the
    single-GPU run bit for bit, dither included, whatever the world size.  No communication."""
    idx = shard_utterances([np.asarray(x).size for x in xs], world_size)[rank]
    return idx, plan.run_host([xs[i] for i in idx], chunk_bytes=chunk_bytes, ids=idx)


def shard_utterances(n_samples, world_size):
    """Split a ragged list of utterances over `world_size` ranks (one process per GPU) with no data exchange:
    longest-processing-time greedy on the sample counts (cost of an utterance ~ 4 N + 4 W D, proportional to N).
    Returns a list of index arrays, one per rank; every utterance appears in exactly one of them.  A single
    utterance is never split across devices."""
    n = np.asarray(n_samples, dtype=np.int64)
    order = np.argsort(-n, kind="stable")
    load = np.zeros(world_size, dtype=np.int64)
    parts = [[] for _ in range(world_size)]
    for i in order:
        r = int(np.argmin(load))
        parts[r].append(int(i))
        load[r] += int(n[i])
    return [np.asarray(sorted(p), dtype=np.int64) for p in parts]
