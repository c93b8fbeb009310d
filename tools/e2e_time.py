"""End-to-end (host memory -> features in host memory) rates of one GPU: the drop-in call per utterance against
speech_plan_run_host on the whole list, float and 16-bit PCM input.   python tools/e2e_time.py [nutt] [seconds]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from speech_amd import api

nutt = int(sys.argv[1]) if len(sys.argv) > 1 else 400
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 10.0
n = int(16000 * secs)
rng = np.random.default_rng(1)
xs = [(rng.standard_normal(n) * 1000).astype(np.float32) for _ in range(nutt)]
x16 = [x.astype(np.int16) for x in xs]
audio_s = nutt * secs
for kind, o in (("mfcc", {}), ("fbank", dict(B=80)), ("spectrogram", {})):
    plan = api.Plan(kind, **o)
    f = getattr(api, "kaldi_" + kind)
    kw = dict(B=80) if kind == "fbank" else {}
    f(xs[0], **kw)
    t0 = time.perf_counter()
    for x in xs[:100]:
        f(x, **kw)
    t_drop = (time.perf_counter() - t0) / 100 * nutt
    outs = plan.run_host(xs)
    res = {}
    for name, inp in (("float", xs), ("pcm16", x16)):
        plan.run_host(inp, out=outs)
        t0 = time.perf_counter()
        for _ in range(3):
            plan.run_host(inp, out=outs)
        res[name] = (time.perf_counter() - t0) / 3
    print("%-12s %d x %.0f s: drop-in calls %.2f M audio-s/s | run_host float %.2f M (%.1f GB/s in+out) | run_host pcm16 %.2f M" % (
        kind, nutt, secs, audio_s / t_drop / 1e6, audio_s / res["float"] / 1e6,
        (sum(x.nbytes for x in xs) + sum(y.nbytes for y in outs)) / res["float"] / 1e9, audio_s / res["pcm16"] / 1e6), flush=True)
