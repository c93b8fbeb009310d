"""Print max errors of the tuned and generic HIP kernels against the oracle (exact and float-FFT modes) and float64 truth."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import oracle_lib as O, np_truth as T, signals
from compare import max_err
from speech_amd import api

TRUTH = {"spectrogram": T.spectrogram, "fbank": T.fbank, "mfcc": T.mfcc, "plp": T.plp}
x = signals.utterance(48000, 31337, flat=True)
x2 = signals.utterance(48000, 77)
for name, xx in (("flat", x), ("tonal", x2)):
    X = torch.from_numpy(xx).cuda()
    for kind, o in [("spectrogram", {}), ("fbank", dict(B=23)), ("fbank", dict(B=40, hif=7600.0)), ("fbank", dict(B=80)), ("mfcc", {}),
                    ("mfcc", dict(B=40, C=40, hif=7600.0)), ("plp", {})]:
        _, ref = O.run(kind, xx, **o)
        _, reff = O.run(kind, xx, exact=False, **o)
        tr = TRUTH[kind](xx.astype(np.float64), single=True, **o)
        row = []
        for fg in (0, 1):
            plan = api.Plan(kind, force_generic=fg, **o)
            b = api.Batch(plan, [xx.size])
            Y = torch.zeros(b.total_out, device="cuda")
            b.run(X, Y); torch.cuda.synchronize()
            got = Y.cpu().numpy().reshape(-1, plan.D)
            row.append("%s: vs-oracle %.2e vs-truth %.2e" % ("generic" if fg else "tuned", max_err(kind, o, got, ref), max_err(kind, o, got, tr)))
        print(name, kind, o, "|", " | ".join(row), "| oracle-vs-truth %.2e, oracle(float fft)-vs-truth %.2e" % (max_err(kind, o, ref, tr), max_err(kind, o, reff, tr)))
