"""Double precision: the quad kernel (quad_f64.hip) against the generic kernel on the same inputs, and their speed."""
import itertools, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from speech_amd import api

rng = np.random.default_rng(5)
worst = 0.0
for kind, snip, dc0, raw, use_e, L_ms in itertools.product(("spectrogram", "fbank", "mfcc", "plp"), (0, 1), (0, 1), (0, 1), (0, 1), (25.0, 20.0, 32.0)):
    o = dict(snip=snip, dc0=dc0, raw_energy=raw, use_energy=use_e, frame_length=L_ms)
    a = api.Plan(kind, dtype="f64", **o)
    g = api.Plan(kind, dtype="f64", force_generic=1, **o)
    assert a.tuned and not g.tuned, (kind, o, a.tuned, g.tuned)
    lens = [int(v) for v in rng.integers(600, 9000, size=7)] + [512, 700, 16000 * 3 + 5]
    ba, bg = api.Batch(a, lens), api.Batch(g, lens)
    X = torch.from_numpy(rng.standard_normal(ba.total_samples) * 3000.0).cuda()
    Ya = torch.full((ba.total_out,), float("nan"), device="cuda", dtype=torch.float64)
    Yg = torch.full((bg.total_out,), float("nan"), device="cuda", dtype=torch.float64)
    ba.run(X, Ya); bg.run(X, Yg)
    torch.cuda.synchronize()
    d = (Ya - Yg).abs().max().item()
    assert d == d, ("nan", kind, o)
    worst = max(worst, d)
    if d > 1e-9: print("DIFF", kind, o, d, flush=True)
print("worst |quad - generic| = %.3g" % worst, flush=True)

for kind in ("mfcc", "spectrogram", "fbank", "plp"):
    for fg in (0, 1):
        plan = api.Plan(kind, dtype="f64", snip=1, dc0=1, raw_energy=1, force_generic=fg)
        b = api.Batch(plan, [160000] * 200)
        X = (torch.randn(b.total_samples, device="cuda", dtype=torch.float64) * 1000)
        Y = torch.zeros(b.total_out, device="cuda", dtype=torch.float64)
        for _ in range(2): b.run(X, Y)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5): b.run(X, Y)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
        print(kind, "generic" if fg else "quad", "%.3f ms per 2000 audio-s -> %.2f M audio-s/s" % (dt * 1e3, 2000.0 / dt / 1e6), flush=True)
