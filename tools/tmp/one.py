import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import signals, oracle_lib as O
from test_fuzz import _configs
from speech_amd import api
for kind, o, lens, L, stp in _configs(25, 144):
    if o["sr"] != 32000.0 or kind != "mfcc" or L != 588: continue
    print(kind, o, lens, L, stp)
    xs = [signals.utterance(m, 1000 + 7 * i + 144, sr=o["sr"], flat=True) for i, m in enumerate(lens)]
    X = torch.from_numpy(np.concatenate(xs)).cuda()
    outs = []
    for fg in (0, 1):
        plan = api.Plan(kind, force_generic=fg, **o)
        b = api.Batch(plan, lens); Y = torch.zeros(b.total_out, device="cuda"); b.run(X, Y); torch.cuda.synchronize()
        y = Y.cpu().numpy()
        outs.append([y[int(b.y_off[u]):int(b.y_off[u]) + int(b.n_frames[u]) * plan.D].reshape(-1, plan.D) for u in range(len(lens))])
    for u in range(len(lens)):
        _, ref = O.run(kind, xs[u], **o)
        d = np.abs(outs[0][u] - outs[1][u])
        print("utt", u, "len", lens[u], "frames", outs[0][u].shape, "tuned-generic", d.max(), "at", np.unravel_index(d.argmax(), d.shape), "tuned-oracle", np.abs(outs[0][u] - ref).max(), "generic-oracle", np.abs(outs[1][u] - ref).max())
        if d.max() > 1:
            fr = np.unravel_index(d.argmax(), d.shape)[0]
            print(" tuned  ", outs[0][u][fr][:6]); print(" generic", outs[1][u][fr][:6]); print(" oracle ", ref[fr][:6])
            print(" bad frames:", np.where(d.max(axis=1) > 1e-2)[0])
