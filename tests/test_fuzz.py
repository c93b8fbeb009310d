"""Differential fuzz of the two HIP implementations: random option sets inside the tuned kernels' envelope (8 / 16 / 32 kHz,
frame lengths 129..1024 samples, even shifts; 48 kHz, 1025..2048 samples, any shift; every switch, ragged utterance lengths,
both edge modes) must give the same
features on the tuned and on the generic kernels (force_generic), and pass the parity suite's acceptance rule against the
CPU oracle."""
import numpy as np
import pytest

import np_truth as T
import oracle_lib as O
import signals
from compare import check_parity, max_err

WINDOWS = ("povey", "hamming", "hann", "rectangular", "blackman")


def _configs(n, seed, xw=False):
    """xw: 2048-point sets only (48 kHz, L = 1025..2048, odd and even shifts); otherwise the 256- / 512- / 1024-point mix"""
    rng = np.random.default_rng(seed)
    out = []
    for i in range(n):
        sr = 48000.0 if xw else float(rng.choice([8000.0, 16000.0, 32000.0]))
        # frame length so that L lands in 129..256 (8 kHz), 257..512 (16 kHz) or 513..1024 (32 kHz), in whole samples
        L = int(rng.integers(129, 257)) if sr == 8000.0 else (int(rng.integers(257, 513)) if sr == 16000.0 else int(rng.integers(513, 1025)))
        if xw:
            L = int(rng.integers(1025, 2049))
        fl = (L + 0.5) * 1000.0 / sr
        stp = int(rng.integers(40, 900)) if xw else 2 * int(rng.integers(10, 200))
        fs = (stp + 0.25) * 1000.0 / sr
        kind = str(rng.choice(["spectrogram", "fbank", "mfcc", "plp"]))
        o = dict(sr=sr, fl=fl, fs=fs, snip=int(rng.integers(0, 2)), dc0=int(rng.integers(0, 2)), raw_energy=int(rng.integers(0, 2)),
                 preemph=float(rng.choice([0.0, 0.5, 0.97])), win=str(rng.choice(WINDOWS)))
        nyq = sr / 2
        if kind != "spectrogram":
            B = int(rng.integers(8, 41 if sr == 8000.0 else 65))      # more bins than that and the lowest filters hold a single weak FFT bin: float32 noise, not a difference of the kernels
            o.update(B=B, lof=float(rng.choice([0.0, 20.0, 100.0])), hif=float(nyq - rng.choice([0.0, 200.0, 400.0])), use_energy=int(rng.integers(0, 2)))
        if kind == "fbank":
            o.update(lg=int(rng.integers(0, 2)), amp=int(rng.integers(0, 2)))
        if kind == "mfcc":
            o.update(C=int(rng.integers(2, min(o["B"], 40) + 1)), Q=float(rng.choice([0.0, 22.0])))
        if kind == "plp":
            o.update(B=int(rng.integers(15, 41)), C=13, Q=float(rng.choice([0.0, 22.0])))
        lens = [int(rng.integers(L, (12 if xw else 40) * L)) for _ in range(3)] + [L, L + stp]
        out.append((kind, o, lens, L, stp))
    return out


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [11, 12, 13, 14, 15, 16, 17, 18, 2048, 2049, 2050])
def test_random_option_sets_tuned_equals_generic(seed):
    import torch
    from speech_amd import api
    worst, ntuned = 0.0, 0
    for kind, o, lens, L, stp in _configs(25, seed, xw=seed >= 2048):
        probe = api.Plan(kind, **o)
        tuned = probe.tuned
        probe.close()
        if not tuned:           # outside the envelope after all (e.g. a frame shift whose slices do not fit in LDS): nothing to compare
            continue
        ntuned += 1
        xs = [signals.utterance(n, 1000 + 7 * i + seed, sr=o["sr"], flat=True) for i, n in enumerate(lens)]
        X = torch.from_numpy(np.concatenate(xs)).cuda()
        outs = []
        for fg in (0, 1):
            plan = api.Plan(kind, force_generic=fg, **o)
            assert plan.tuned == (fg == 0), (kind, o)
            assert (plan.L, plan.stp) == (L, stp), (plan.L, plan.stp, L, stp)
            b = api.Batch(plan, lens)
            Y = torch.zeros(b.total_out, device="cuda")
            b.run(X, Y)
            torch.cuda.synchronize()
            y = Y.cpu().numpy()
            outs.append([y[int(b.y_off[u]):int(b.y_off[u]) + int(b.n_frames[u]) * plan.D].reshape(-1, plan.D) for u in range(len(lens))])
            b.close(); plan.close()
        # the suite's error measures (tests/compare.py): weighted log power for the spectrogram, absolute for log-mel and
        # cepstra, relative to the frame's largest bin for linear mel energies; two float32 implementations each within
        # the tolerance of the exact result can be twice as far from each other, weak mel bins a little more
        tol = 5e-5 if kind == "spectrogram" else (2e-5 if (kind == "fbank" and not o["lg"]) else 5e-4)
        # and the acceptance rule of the parity suite against the CPU oracle (exact-FFT and float-FFT) and the float64
        # statement, on the longest of the utterances
        u0 = int(np.argmax(lens))
        x0, got0 = xs[u0], outs[0][u0]
        _, ref = O.run(kind, x0, single=True, **o)
        _, ref_fast = O.run(kind, x0, single=True, exact=False, **o)
        truth = {"spectrogram": T.spectrogram, "fbank": T.fbank, "mfcc": T.mfcc, "plp": T.plp}[kind](np.asarray(x0, dtype=np.float64), single=True, **o)
        # the suite's one rule, the suite's one set of margins (compare.MARGIN, MAX_MARGIN, CAP_FACTORS): no per-set widening
        ok, e_go, e_gt, e_ot = check_parity(kind, o, got0, ref, truth, tol=1e-5 if kind == "spectrogram" else 1e-4, oracle_fast=ref_fast)
        assert ok, (kind, str(o), lens[u0], "hip-vs-oracle %.3g, hip-vs-f64 %.3g, f32-cpu-vs-f64 %.3g" % (e_go, e_gt, e_ot))
        for u, (t, g) in enumerate(zip(*outs)):
            if t.shape[0] == 0:
                continue
            assert np.isfinite(t).all() and np.isfinite(g).all(), (kind, o, u)
            err = float(max_err(kind, o, t, g))
            worst = max(worst, err / tol)
            assert err <= tol, (kind, o, lens, u, err)
    assert ntuned >= 12 and worst > 0.0      # most sets are tuned; the two implementations are not the same code
