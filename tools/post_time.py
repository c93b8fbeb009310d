"""Time the corpus post-processing passes (post_kernels.hip) on a packed feature batch and report the HBM rate of each:
python tools/post_time.py [kind] [nutt] [seconds]      (default fbank B=80, 1000 utterances of 10 s)"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from speech_amd import api

kind = sys.argv[1] if len(sys.argv) > 1 else "fbank"
nutt = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
secs = float(sys.argv[3]) if len(sys.argv) > 3 else 10.0
opts = dict(B=80) if kind == "fbank" else {}
plan = api.Plan(kind, **opts)
lens = [int(16000 * secs)] * nutt
b = api.Batch(plan, lens)
D = plan.D
Y = torch.randn(b.total_out, device="cuda")
nel = b.total_out


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


rows = []
Z1 = Y.clone()
rows.append(("cmvn (moments + affine)", timed(lambda: b.cmvn(Z1, norm_vars=True)), 3 * 4 * nel))         # read, read, write
sc, of = np.ones(D), np.zeros(D)
rows.append(("affine", timed(lambda: b.affine(Z1, sc, of)), 2 * 4 * nel))
rows.append(("stats", timed(lambda: b.stats(Y)), 4 * nel))
Z3 = torch.empty(nel * 3, device="cuda")
rows.append(("add_deltas order 2 window 2", timed(lambda: b.add_deltas(Y, Z3, 2, 2)), 4 * 4 * nel))
Z9 = torch.empty(nel * 9, device="cuda")
rows.append(("splice 4+1+4", timed(lambda: b.splice(Y, Z9, 4, 4)), 10 * 4 * nel))
_, tot = b.subsample_rows(3, 0)
Z7 = torch.empty(tot * D * 7, device="cuda")
rows.append(("splice 3+1+3, every 3rd frame", timed(lambda: b.splice_subsample(Y, Z7, 3, 3, 3, 0)), 4 * nel + 4 * tot * D * 7))
Zs = torch.empty(tot * D, device="cuda")
rows.append(("subsample 3", timed(lambda: b.splice_subsample(Y, Zs, 0, 0, 3, 0)), 4 * tot * D + 4 * tot * D))   # kept rows read, written
print("%s D=%d, %d utterances x %.0f s: %d frames, %.1f MB of features" % (kind, D, nutt, secs, b.total_frames, 4e-6 * nel))
for name, ms, byts in rows:
    print("%-32s %8.3f ms  %7.2f TB/s (algorithmic bytes %.0f MB)" % (name, ms, byts / ms * 1e-9, byts * 1e-6))
