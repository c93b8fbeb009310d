"""2048-point (xwide) plans: tuned kernels against the generic kernels on the same audio (run on the GPU box)."""
import sys, os
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import signals
from speech_amd import api

cases = [("spectrogram", dict(sr=44100.0)), ("spectrogram", dict(sr=48000.0, snip=0)), ("fbank", dict(sr=48000.0, hif=24000.0, B=80)),
         ("mfcc", dict(sr=44100.0, hif=20000.0, B=40, C=20)), ("mfcc", dict(sr=48000.0, hif=24000.0, snip=0)), ("plp", dict(sr=48000.0, hif=24000.0)),
         ("fbank", dict(sr=16000.0, fl=128.0, fs=32.0, B=23)), ("mfcc", dict(sr=16000.0, fl=64.0625, fs=10.0)),
         ("mfcc", dict(sr=48000.0, hif=24000.0, dither=2.0)), ("spectrogram", dict(sr=44100.0, dither=2.0, snip=0))]
lens = [50000, 9001, 2500]
for kind, o in cases:
    sr = o.get("sr", 16000.0)
    xs = [signals.utterance(n, 40 + i, sr=sr) for i, n in enumerate(lens)]
    if not o.get("snip", 1):
        pass
    X = torch.from_numpy(np.concatenate(xs)).cuda()
    outs = []
    for fg in (0, 1):
        plan = api.Plan(kind, force_generic=fg, dither_seed=3, **o)
        b = api.Batch(plan, lens)
        Y = torch.zeros(b.total_out, device="cuda")
        b.run(X, Y)
        torch.cuda.synchronize()
        outs.append((plan.tuned, Y.cpu().numpy().reshape(-1, plan.D)))
    d = np.abs(outs[0][1] - outs[1][1])
    print(kind, o, "tuned", outs[0][0], outs[1][0], "max diff", d.max(), "at", np.unravel_index(d.argmax(), d.shape), "finite", np.isfinite(outs[0][1]).all(), flush=True)
