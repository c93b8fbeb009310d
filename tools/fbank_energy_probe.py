"""Probe of the two fbank sets the final 1 500-set oracle sweep flagged (use_energy = 1, raw_energy = 0, frames of 130 / 168 samples):
is the HIP output's noise level systematically above the CPU arithmetic's, and in which columns?   python tools/fbank_energy_probe.py"""
import os, sys, numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import signals, oracle_lib as O, np_truth as T
from speech_amd import api
SETS = [dict(sr=16000.0, fl=10.498742975412526, fs=19.007771014104765, snip=0, dc0=1, raw_energy=0, preemph=0.97, win="povey", B=40, lof=100.0, hif=8000.0, use_energy=1, lg=1, amp=0),
        dict(sr=22050.0, fl=5.875928096669526, fs=7.482799407003186, snip=1, dc0=1, raw_energy=0, preemph=0.97, win="povey", B=54, lof=20.0, hif=10725.0, use_energy=1, lg=1, amp=1)]
for o in SETS:
    for var in ("as flagged", "raw_energy=1", "use_energy=0"):
        oo = dict(o)
        if var == "raw_energy=1": oo["raw_energy"] = 1
        if var == "use_energy=0": oo["use_energy"] = 0
        kw = {k: v for k, v in oo.items() if k not in ("fl", "fs", "snip", "win")}
        kw.update(frame_length=oo["fl"], frame_shift=oo["fs"], snip_edges=oo["snip"], win_type=oo["win"])
        rat, rat_e, rat_m = [], [], []
        for seed in range(6):
            x = signals.utterance(int(oo["sr"] * 2.0), 900 + seed, sr=oo["sr"], dtype=np.float32)
            got = api.kaldi_fbank(x, **kw)
            ref = O.run("fbank", x, single=True, **oo)[1]; fast = O.run("fbank", x, single=True, exact=False, **oo)[1]
            tr = T.fbank(np.asarray(x, dtype=np.float64), single=True, **oo)
            rms = lambda a, sl: float(np.sqrt(np.mean((a[:, sl] - tr[:, sl]) ** 2)))
            e = 1 if oo["use_energy"] else 0
            allc, en, mel = slice(None), slice(0, e), slice(e, None)
            rat.append(rms(got, allc) / max(rms(ref, allc), rms(fast, allc)))
            if e: rat_e.append(rms(got, en) / max(rms(ref, en), rms(fast, en), 1e-30))
            rat_m.append(rms(got, mel) / max(rms(ref, mel), rms(fast, mel)))
        print("sr %.0f L %d %s tuned %s: rms ratio all columns %s | energy column %s | mel columns %s" % (oo["sr"], int(np.float32(oo["sr"]) * np.float32(oo["fl"]) / np.float32(1000.0)), var,
              api.Plan("fbank", **{k: v for k, v in oo.items()}).tuned, " ".join("%.2f" % r for r in rat), " ".join("%.2f" % r for r in rat_e), " ".join("%.2f" % r for r in rat_m)), flush=True)
