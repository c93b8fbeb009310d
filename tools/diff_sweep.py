"""Extended differential sweep of the tuned against the generic kernels (the option generator of tests/test_fuzz.py over many
more seeds, each set with and without dither): python tools/diff_sweep.py [first_seed] [n_seeds] [xw]   (xw: 2048-point sets only).  Prints the worst distance
per feature kind; a defect of one implementation shows up as a distance far above 1e-3 (round 1: mfcc with use_energy and C == 2)."""
import os, sys, numpy as np, torch
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import signals
from compare import max_err
from test_fuzz import _configs
from speech_amd import api
worst = {}
n = 0
S0 = int(sys.argv[1]) if len(sys.argv) > 1 else 100
NS = int(sys.argv[2]) if len(sys.argv) > 2 else 40
XW = len(sys.argv) > 3 and sys.argv[3] == "xw"
for seed in range(S0, S0 + NS):
    for kind, o, lens, L, stp in _configs(25, seed, xw=XW):
        for dither in (0.0, 2.0):
            oo = dict(o, dither=dither, dither_seed=seed)
            p0 = api.Plan(kind, **oo)
            if not p0.tuned: continue
            xs = [signals.utterance(m, 1000 + 7 * i + seed, sr=o["sr"], flat=True) for i, m in enumerate(lens)]
            X = torch.from_numpy(np.concatenate(xs)).cuda()
            outs = []
            for fg in (0, 1):
                plan = api.Plan(kind, force_generic=fg, **oo)
                b = api.Batch(plan, lens); Y = torch.zeros(b.total_out, device="cuda"); b.run(X, Y); torch.cuda.synchronize()
                outs.append(Y.cpu().numpy().reshape(-1, plan.D))
            assert np.isfinite(outs[0]).all(), (seed, kind, oo)
            e = max_err(kind, o, outs[0], outs[1])
            key = (kind, o.get("lg", 1), dither > 0)
            if e > worst.get(key, (0,))[0]: worst[key] = (e, seed, L, stp, o.get("B"))
            n += 1
print("configs", n)
for k, v in sorted(worst.items(), key=str): print(k, v)
