"""Random sweep of the corpus post-processing passes (splice, splice + subsample, add-deltas, statistics, CMVN) against
oracle/post_oracle.py: feature widths 2-513, ragged batches, offsets and array starts that are / are not 16-byte aligned, f32 and
f64.   python tools/post_sweep.py [seed] [n_sets]   (round 1: 1 800 sets, no miss)"""
import os, sys, numpy as np, torch
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import post_oracle as PO
from speech_amd import api
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
bad = 0
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 150):
    dtype = "f64" if rng.integers(0, 3) == 0 else "f32"
    B = int(rng.integers(2, 129)); ue = int(rng.integers(0, 2))
    if rng.integers(0, 6) == 0:
        plan = api.Plan("spectrogram", dtype=dtype, sr=float(rng.choice([8000.0, 16000.0])), snip_edges=0)
    else:
        plan = api.Plan("fbank", dtype=dtype, B=B, use_energy=ue, snip_edges=0)
    D = plan.D
    nutt = int(rng.integers(1, 12))
    lens = [int(rng.integers(80, 6000)) for _ in range(nutt)]
    W = [plan.num_frames(n) for n in lens]
    gap = int(rng.choice([0, 0, 1, 2, 3, 4]))
    y_off = np.cumsum([0] + [w * D + gap for w in W[:-1]]).astype(np.uint64)
    b = api.Batch(plan, lens, y_off=y_off)
    td = torch.float64 if dtype == "f64" else torch.float32
    nel = int(y_off[-1]) + W[-1] * D
    shift = int(rng.integers(0, 2))
    Y = (torch.randn(nel + 4, device="cuda", dtype=td) * 5)[shift:shift + nel]
    y0 = Y.cpu().numpy()
    utts = [y0[int(y_off[u]):int(y_off[u]) + W[u] * D].reshape(-1, D) for u in range(nutt)]
    # splice
    left, right = int(rng.integers(0, 6)), int(rng.integers(0, 6)); K = left + right + 1
    Z = torch.zeros(nel * K + 4, device="cuda", dtype=td)[shift:]
    b.splice(Y, Z, left, right); torch.cuda.synchronize(); z = Z.cpu().numpy()
    for u in range(nutt):
        got = z[int(y_off[u]) * K:int(y_off[u]) * K + W[u] * D * K].reshape(-1, D * K)
        if not np.array_equal(got, PO.splice(utts[u], left, right)): bad += 1; print("SPLICE", dtype, D, lens, left, right, gap, shift, u); break
    # splice + subsample
    n = int(rng.integers(1, 6)); off = int(rng.integers(0, n))
    rows, tot = b.subsample_rows(n, off)
    Z = torch.full((max(1, tot * D * K) + 4,), 7.0, device="cuda", dtype=td)[shift:]
    b.splice_subsample(Y, Z, left, right, n, off); torch.cuda.synchronize(); z = Z.cpu().numpy(); r0 = 0
    for u in range(nutt):
        want = PO.subsample(PO.splice(utts[u], left, right), n, off)
        got = z[r0 * D * K:(r0 + int(rows[u])) * D * K].reshape(-1, D * K)
        if want.shape != got.shape or not np.array_equal(got, want): bad += 1; print("SUBSAMPLE", dtype, D, lens, left, right, n, off, gap, shift, u); break
        r0 += int(rows[u])
    # deltas
    order, window = int(rng.integers(0, 4)), int(rng.integers(1, 6))
    if rng.integers(0, 2): order, window = 2, 2
    Z = torch.zeros(nel * (order + 1) + 4, device="cuda", dtype=td)[shift:]
    b.add_deltas(Y, Z, order, window); torch.cuda.synchronize(); z = Z.cpu().numpy()
    for u in range(nutt):
        got = z[int(y_off[u]) * (order + 1):int(y_off[u]) * (order + 1) + W[u] * D * (order + 1)].reshape(-1, D * (order + 1))
        want = PO.add_deltas(utts[u], order, window)
        if np.abs(got - want).max() > (1e-12 if dtype == "f64" else 1e-5) * max(1.0, np.abs(want).max()): bad += 1; print("DELTAS", dtype, D, lens, order, window, gap, shift, u, np.abs(got - want).max()); break
    # cmvn + stats
    s, q, nf = b.stats(Y)
    allf = np.concatenate(utts).astype(np.float64)
    if nf != allf.shape[0] or not np.allclose(s, allf.sum(0), rtol=1e-10, atol=1e-7) or not np.allclose(q, (allf * allf).sum(0), rtol=1e-10, atol=1e-7): bad += 1; print("STATS", dtype, D, lens)
    Zc = Y.clone(); nv = bool(rng.integers(0, 2)); b.cmvn(Zc, norm_vars=nv); torch.cuda.synchronize(); zc = Zc.cpu().numpy()
    for u in range(nutt):
        got = zc[int(y_off[u]):int(y_off[u]) + W[u] * D].reshape(-1, D); want = PO.cmvn(utts[u], nv)
        if W[u] > 1 and np.abs(got - want).max() > (1e-9 if dtype == "f64" else 3e-4) * max(1.0, np.abs(want).max()): bad += 1; print("CMVN", dtype, D, lens, nv, u, np.abs(got - want).max()); break
    b.close(); plan.close()
print("bad", bad)
