"""Independent float64 numpy statement of the reference algorithm (test infrastructure).

Written separately from oracle/ (vectorised over frames, numpy's rfft, dense matrices) so that
an indexing or ordering mistake in the C restatement does not silently repeat here.  It follows
SURVEY.md section 3.3 / 8(a) and cites the same reference lines:
  framing c/kaldi_mfcc.c:216-240, DC removal :265-271, raw energy :274-279/:295-301,
  pre-emphasis :282-288, window :105-135/:291, power :307-310, mel table :157-188,
  DCT/lifter :190-210,:323-333, spectrogram epilogue c/kaldi_spectrogram.c:236-251,
  fbank epilogue c/kaldi_fbank.c:290-350, PLP per SURVEY.md 8(a) row P.
Everything is float64; L and stp are taken in the precision the entry point would use.
"""
import numpy as np

FLT_EPS = float(np.finfo(np.float32).eps)


def sizes(N, sr, fl, fs, snip, single=True):
    if single:
        L = int(np.float32(sr) * np.float32(fl) / np.float32(1000.0))
        stp = int(np.float32(sr) * np.float32(fs) / np.float32(1000.0))
    else:
        L = int(float(sr) * float(fl) / 1000.0)
        stp = int(float(sr) * float(fs) / 1000.0)
    W = 1 + (N - L) // stp if snip else (N + stp // 2) // stp
    nfft = 1
    while nfft < L:
        nfft *= 2
    return L, stp, W, nfft


def frames(x, L, stp, W, snip):
    N = len(x)
    if snip:
        idx = np.arange(W)[:, None] * stp + np.arange(L)[None, :]
    else:
        ss = stp // 2 - L // 2
        idx = ss + np.arange(W)[:, None] * stp + np.arange(L)[None, :]
        for _ in range(64):  # reflect until in range
            neg = idx < 0
            idx = np.where(neg, -idx - 1, idx)
            big = idx >= N
            idx = np.where(big, 2 * N - 1 - idx, idx)
            if not (neg.any() or big.any()):
                break
    return np.asarray(x, dtype=np.float64)[idx]


def window(L, kind):
    l = np.arange(L, dtype=np.float64)
    a = 2.0 * np.pi / (L - 1)
    if kind.startswith("rectangular"):
        return np.ones(L)
    if kind.startswith("blackman"):
        return 0.42 - 0.5 * np.cos(l * a) + 0.58 * np.cos(2 * l * a)   # sic: coeff1 = 1-0.42
    if kind.startswith("hamming"):
        return 0.54 - 0.46 * np.cos(l * a)
    if kind.startswith("hann"):
        return 0.5 - 0.5 * np.cos(l * a)
    if kind.startswith("povey"):
        return (0.5 - 0.5 * np.cos(l * a)) ** 0.85
    raise ValueError(kind)


MEAN_HOOK = None      # tools/mean_noise.py: a function frames (W x L, float64) -> W x 1 means, to measure what a float32 mean costs


def front_end(x, sr, fl, fs, snip, dc0, preemph, win, single=True, energy_before_dc=False):
    """-> (windowed frames W x L, pre-energy W, post-energy W, nfft)"""
    L, stp, W, nfft = sizes(len(x), sr, fl, fs, snip, single)
    fr = frames(x, L, stp, W, snip)
    e_raw0 = (fr * fr).sum(axis=1)
    if dc0:
        fr = fr - (MEAN_HOOK(fr) if MEAN_HOOK is not None else fr.mean(axis=1, keepdims=True))
    e_pre = e_raw0 if energy_before_dc else (fr * fr).sum(axis=1)
    if L >= 2 and preemph >= FLT_EPS:
        out = fr.copy()
        out[:, 1:] = fr[:, 1:] - preemph * fr[:, :-1]
        out[:, 0] = fr[:, 0] - preemph * fr[:, 0]
        fr = out
    fr = fr * window(L, win)[None, :]
    e_post = (fr * fr).sum(axis=1)
    return fr, np.maximum(e_pre, FLT_EPS), np.maximum(e_post, FLT_EPS), nfft


def power(fr, nfft):
    Z = np.fft.rfft(fr, n=nfft, axis=1)
    return Z.real ** 2 + Z.imag ** 2


def mel_matrix(sr, nfft, B, lof, hif):
    """dense F x B triangular weights, last bin's right edge = himel (c/kaldi_mfcc.c:176-187)"""
    F = nfft // 2 + 1
    mel = lambda f: 1127.0 * np.log(1.0 + np.asarray(f, dtype=np.float64) / 700.0)
    mels = mel(np.arange(F) * (sr / nfft))
    lomel, himel = mel(lof), mel(hif)
    dmel = (himel - lomel) / (B + 1)
    M = np.zeros((F, B))
    for b in range(B):
        l = lomel + b * dmel
        c = l + dmel
        r = himel if (b == B - 1 and B >= 2) else c + dmel
        up = (mels >= l) & (mels < c)
        dn = (mels >= c) & (mels < r)
        M[up, b] = (mels[up] - l) / dmel
        M[dn, b] = (r - mels[dn]) / dmel
    return M, lomel, dmel


def spectrogram(x, sr=16000.0, fl=25.0, fs=10.0, snip=1, dc0=1, raw_energy=1, preemph=0.97, win="povey", single=True):
    fr, e_pre, e_post, nfft = front_end(x, sr, fl, fs, snip, dc0, preemph, win, single)
    P = power(fr, nfft)
    Y = P.copy()
    Y[:, 0] = e_pre if raw_energy else e_post
    Y[:, -1] = Y[:, -2]
    return np.log(np.maximum(Y, FLT_EPS))


def fbank(x, sr=16000.0, fl=25.0, fs=10.0, snip=1, dc0=1, raw_energy=1, preemph=0.97, win="povey",
          amp=0, lof=20.0, hif=8000.0, B=23, lg=1, use_energy=0, single=True):
    fr, e_pre, e_post, nfft = front_end(x, sr, fl, fs, snip, dc0, preemph, win, single)
    P = power(fr, nfft)
    if amp:
        P = np.sqrt(P)
    M, _, _ = mel_matrix(sr, nfft, B, lof, hif)
    Y = P @ M
    if lg:
        Y = np.log(np.maximum(Y, FLT_EPS))
    if B >= 2:
        Y[:, -1] = Y[:, -2]
    if use_energy:
        e = np.log(e_pre if raw_energy else e_post)
        if B == 1:
            Y[:, 0] = e
        Y = np.concatenate([e[:, None], Y], axis=1)
    return Y


def mfcc(x, sr=16000.0, fl=25.0, fs=10.0, snip=1, dc0=1, raw_energy=1, preemph=0.97, win="povey",
         lof=20.0, hif=8000.0, B=23, C=13, Q=22.0, use_energy=0, single=True):
    fr, e_pre, e_post, nfft = front_end(x, sr, fl, fs, snip, dc0, preemph, win, single,
                                        energy_before_dc=(not single))   # c/kaldi_mfcc.c:576-591
    P = power(fr, nfft)
    M, _, _ = mel_matrix(sr, nfft, B, lof, hif)
    lm = np.log(np.maximum(P @ M, FLT_EPS))
    n = np.arange(B)
    k = np.arange(C)
    dct = 2.0 * np.cos(np.pi * (n[None, :] + 0.5) * k[:, None] / B)           # REDFT10
    lift = (1.0 + 0.5 * Q * np.sin(np.pi * k / Q)) if Q > FLT_EPS else np.ones(C)
    lift = lift * (1.0 / np.sqrt(2.0 * B))
    lift[0] = (1.0 if Q <= FLT_EPS else 1.0) * 0.5 / np.sqrt(B)
    Y = (lm @ dct.T) * lift[None, :]
    if use_energy:
        Y[:, 0] = np.log(e_pre if raw_energy else e_post)
    if C >= 2:
        Y[:, -1] = Y[:, -2]
    return Y


def plp(x, sr=16000.0, fl=25.0, fs=10.0, snip=1, dc0=1, raw_energy=1, preemph=0.97, win="povey",
        lof=20.0, hif=8000.0, B=23, compress=1.0 / 3.0, C=13, Q=22.0, cep_scale=1.0, use_energy=0, order=None, single=True):
    order = C - 1 if not order else order
    fr, e_pre, e_post, nfft = front_end(x, sr, fl, fs, snip, dc0, preemph, win, single)
    P = power(fr, nfft)
    M, lomel, dmel = mel_matrix(sr, nfft, B, lof, hif)
    me = P @ M
    cf = 700.0 * (np.exp((lomel + (np.arange(B) + 1) * dmel) / 1127.0) - 1.0)
    fsq = cf * cf
    eql = (fsq / (fsq + 1.6e5)) ** 2 * ((fsq + 1.44e6) / (fsq + 9.61e6))
    v = (me * eql[None, :]) ** compress
    d = np.concatenate([v[:, :1], v, v[:, -1:]], axis=1)                      # B+2
    i = np.arange(order + 1)[:, None]
    j = np.arange(B + 2)[None, :]
    scale = 1.0 / (2.0 * (B + 1))
    bases = 2.0 * scale * np.cos(np.pi / (B + 1) * i * j)
    bases[:, 0] = scale
    bases[:, B + 1] = scale * np.cos(np.pi / (B + 1) * i[:, 0] * (B + 1))
    ac = d @ bases.T                                                          # W x (order+1)
    W = ac.shape[0]
    Y = np.zeros((W, C))
    for w in range(W):
        r = ac[w]
        a = np.zeros(order)
        E = r[0]
        for m in range(order):
            ki = (r[m + 1] + np.dot(a[:m], r[m:0:-1])) / E
            E *= max(1.0 - ki * ki, 1e-5)
            new = a.copy()
            new[m] = -ki
            new[:m] = a[:m] - ki * a[:m][::-1]
            a = new
        c = np.zeros(order)
        for m in range(order):
            s = sum((m - q) * a[q] * c[m - q - 1] for q in range(m))
            c[m] = -a[m] - s / (m + 1)
        Y[w, 0] = np.log(max(E, FLT_EPS))
        Y[w, 1:C] = c[:C - 1]
    k = np.arange(C)
    lift = (1.0 + 0.5 * Q * np.sin(np.pi * k / Q)) if Q > FLT_EPS else np.ones(C)
    Y = Y * lift[None, :] * cep_scale
    if use_energy:
        Y[:, 0] = np.log(e_pre if raw_energy else e_post)
    if C >= 2:
        Y[:, -1] = Y[:, -2]
    return Y


def mean_norm(Y):
    return Y - Y.mean(axis=0, keepdims=True)
