"""Random sweep of the drop-in entry points against the CPU oracle and the float64 numpy statement (tests/ infrastructure):
sample rates 4-48 kHz, frame lengths 4-40 ms, any shift (tuned and generic kernels, f32 and f64), every switch.
python tools/oracle_sweep.py [seed] [n_sets].  Round 1: 2 900 sets, the only misses two PLP sets of LPC order 18-19 without
compression, where float32 Levinson-Durbin diverges in the oracle as well (ill-conditioned, not a kernel defect)."""
import os, sys, numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import signals, oracle_lib as O, np_truth as T
from compare import check_parity
import compare
from speech_amd import api
WINDOWS = ("povey", "hamming", "hann", "rectangular", "blackman")
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
bad = 0; n = 0; ntuned = 0
worst = 0
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 300):
    sr = float(rng.choice([4000.0, 8000.0, 11025.0, 16000.0, 22050.0, 32000.0, 44100.0, 48000.0]))
    fl = float(rng.uniform(4.0, 40.0)); fs = float(rng.uniform(2.0, 20.0))
    kind = str(rng.choice(["spectrogram", "fbank", "mfcc", "plp"]))
    single = bool(rng.integers(0, 4) > 0)
    o = dict(sr=sr, fl=fl, fs=fs, snip=int(rng.integers(0, 2)), dc0=int(rng.integers(0, 2)), raw_energy=int(rng.integers(0, 2)),
             preemph=float(rng.choice([0.0, 0.5, 0.97])), win=str(rng.choice(WINDOWS)), mn0=int(rng.integers(0, 4) == 0))
    if kind == "spectrogram": o["mn0"] = 0
    if kind != "spectrogram":
        B = int(rng.integers(2, 60))
        o.update(B=B, lof=float(rng.choice([0.0, 20.0, 100.0])), hif=float(sr / 2 - rng.choice([0.0, 100.0, 300.0])), use_energy=int(rng.integers(0, 2)))
    if kind == "fbank": o.update(lg=int(rng.integers(0, 2)), amp=int(rng.integers(0, 2)))
    if kind == "mfcc": o.update(C=int(rng.integers(1, o["B"] + 1)), Q=float(rng.choice([0.0, 22.0])))
    if kind == "plp":
        o.update(B=int(rng.integers(8, 41)), Q=float(rng.choice([0.0, 22.0])), compress=float(rng.choice([1.0 / 3.0, 0.5, 1.0])))
        o["C"] = int(rng.integers(2, min(o["B"], 20) + 1))
    if kind in ("fbank",) and o.get("mn0") and o.get("use_energy"): o["mn0"] = 0      # reference's mn0 is only defined for D == B
    if kind in ("mfcc", "plp") and o.get("mn0"): o["mn0"] = 0
    L = int(np.float32(sr) * np.float32(fl) / np.float32(1000.0)) if single else int(sr * fl / 1000.0)
    N = int(rng.integers(max(L, 3), 30 * max(L, 3)))
    x = signals.utterance(N, 5000 + it, sr=sr, dtype=np.float32 if single else np.float64)
    rc, ref = O.run(kind, x, single=single, **o)
    fn = getattr(api, "kaldi_" + kind)
    kw = {k: v for k, v in o.items() if k not in ("fl", "fs", "snip", "win")}
    kw.update(frame_length=fl, frame_shift=fs, snip_edges=o["snip"], win_type=o["win"])
    try:
        got = fn(x, **kw)
        err = None
    except api.SpeechError as e:
        got, err = None, str(e)
    n += 1
    if rc != 0 or ref is None:
        if got is not None and got.shape[0] > 0:
            bad += 1; print("ORACLE REJECTS, HIP ACCEPTS", kind, o, N, flush=True)
        continue
    if got is None:
        bad += 1; print("HIP REJECTS", kind, o, N, err, flush=True); continue
    if got.shape != ref.shape:
        bad += 1; print("SHAPE", kind, o, N, got.shape, ref.shape, flush=True); continue
    truth = {"spectrogram": T.spectrogram, "fbank": T.fbank, "mfcc": T.mfcc, "plp": T.plp}[kind](np.asarray(x, dtype=np.float64), single=single, **{k: v for k, v in o.items() if k != "mn0"})
    if o.get("mn0"): truth = T.mean_norm(truth)
    _, ref_fast = O.run(kind, x, single=single, exact=False, **o)
    ok, e_go, e_gt, e_ot = check_parity(kind, o, got, ref, truth, tol=(1e-5 if kind == "spectrogram" else 1e-4) if single else 1e-9, oracle_fast=ref_fast)
    if not ok or not np.isfinite(got).all():
        bad += 1; print("PARITY", kind, "single" if single else "double", o, N, "hip-oracle %.3g hip-f64 %.3g oracle-f64 %.3g rms ratio %.3g tuned %s" % (e_go, e_gt, e_ot, compare.PARITY_LOG[-1][6], api.Plan(kind, **{k: v for k, v in o.items() if k != "mn0"}).tuned if single else False), flush=True)
print("sets", n, "bad", bad)
