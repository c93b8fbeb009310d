"""Timing of the generic kernels (force_generic f32, and f64) on 200 x 10 s utterances."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from speech_amd import api
for kind, dtype, o in (("mfcc", "f32", dict(force_generic=1)), ("mfcc", "f64", {}), ("spectrogram", "f64", {}), ("fbank", "f32", dict(sr=22050.0, hif=11025.0, B=40)),
                       ("fbank", "f32", dict(sr=48000.0, hif=24000.0, B=80, force_generic=1)), ("mfcc", "f32", dict(sr=44100.0, hif=22050.0, force_generic=1)),
                       ("spectrogram", "f32", dict(sr=48000.0, force_generic=1))):
    plan = api.Plan(kind, dtype=dtype, snip=1, dc0=1, raw_energy=1, **o)
    n = int(10 * o.get("sr", 16000.0))
    b = api.Batch(plan, [n] * 200)
    tdt = torch.float64 if dtype == "f64" else torch.float32
    X = (torch.randn(b.total_samples, device="cuda") * 1000).to(tdt)
    Y = torch.zeros(b.total_out, device="cuda", dtype=tdt)
    for _ in range(2): b.run(X, Y)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): b.run(X, Y)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    print(kind, dtype, o, "%.3f ms per 2000 audio-s -> %.2f M audio-s/s" % (dt * 1e3, 2000.0 / dt / 1e6), flush=True)
