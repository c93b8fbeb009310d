"""Corpus post-processing: CMVN, corpus-wide statistics, deltas, splicing -- HIP kernels through the
C ABI against oracle/post_oracle.py (numpy; parity unpinned, see its header), and the rank-merge of statistics over gloo."""
import os
import subprocess
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "oracle"))
import post_oracle as PO  # noqa: E402
import signals  # noqa: E402

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def test_delta_filters_match_the_published_ones():
    f = PO.delta_filters(2, 2)
    assert np.allclose(f[1], np.array([-2, -1, 0, 1, 2]) / 10.0)
    assert np.allclose(f[2], np.array([4, 4, 1, -4, -10, -4, 1, 4, 4]) / 100.0)
    # a linear ramp has slope 1 in the interior, and delta-delta 0
    y = np.arange(40, dtype=np.float32)[:, None] * np.ones((1, 3), np.float32)
    z = PO.add_deltas(y, 2, 2)
    assert np.allclose(z[4:-4, 3:6], 1.0) and np.allclose(z[4:-4, 6:9], 0.0)


def test_oracle_cmvn_and_splice_properties():
    rng = np.random.default_rng(0)
    y = (rng.standard_normal((50, 7)) * 3 + 5).astype(np.float32)
    z = PO.cmvn(y, norm_vars=True)
    assert np.abs(z.mean(axis=0)).max() < 1e-5 and np.abs(z.std(axis=0) - 1).max() < 1e-5
    s = PO.splice(y, 2, 1)
    assert s.shape == (50, 28) and np.array_equal(s[:, 14:21], y) and np.array_equal(s[0, :7], y[0]) and np.array_equal(s[-1, 21:], y[-1])


def test_merge_stats_over_two_ranks_gloo(tmp_path):
    """the corpus-wide statistics are one all-reduce of 2 D + 1 doubles: two CPU ranks, gloo"""
    script = tmp_path / "w.py"
    script.write_text('''
import os, sys, numpy as np, torch.distributed as dist
sys.path.insert(0, %r)
os.environ["SPEECH_AMD_NO_TORCH"] = "1"
from speech_amd import api
dist.init_process_group("gloo")
r = dist.get_rank()
rng = np.random.default_rng(r)
y = rng.standard_normal((30 + 10 * r, 5)) + r
s, q, n = api.merge_stats(y.sum(0), (y * y).sum(0), y.shape[0])
ys = [np.random.default_rng(k).standard_normal((30 + 10 * k, 5)) + k for k in range(2)]
full = np.concatenate(ys)
assert n == full.shape[0] and np.allclose(s, full.sum(0)) and np.allclose(q, (full * full).sum(0))
scale, offset = api.cmvn_from_stats(s, q, n)
z = full * scale + offset
assert np.abs(z.mean(0)).max() < 1e-12 and np.abs(z.std(0) - 1).max() < 1e-12
dist.destroy_process_group()
print("ok", r)
''' % ROOT)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                        "--master-port", "29533", str(script)], capture_output=True, env=env, timeout=300)
    assert r.returncode == 0, r.stderr.decode()[-2000:]


def _run_batch(kind, lens, dtype="f32", **o):
    import torch
    from speech_amd import api
    plan = api.Plan(kind, dtype=dtype, **o)
    b = api.Batch(plan, lens)
    xs = [signals.utterance(n, 300 + i) for i, n in enumerate(lens)]
    tdt = torch.float64 if dtype == "f64" else torch.float32
    X = torch.from_numpy(np.concatenate(xs)).to(tdt).cuda()
    Y = torch.zeros(b.total_out, device="cuda", dtype=tdt)
    b.run(X, Y)
    torch.cuda.synchronize()
    return plan, b, Y


def _utts(b, arr, K=1):
    D = b.plan.D * K
    return [arr[int(b.y_off[u]) * K:int(b.y_off[u]) * K + int(b.n_frames[u]) * D].reshape(-1, D) for u in range(b.nutt)]


@pytest.mark.gpu
@pytest.mark.parametrize("kind,o,dtype", [("mfcc", {}, "f32"), ("fbank", dict(B=80), "f32"), ("spectrogram", {}, "f32"), ("mfcc", {}, "f64")])
def test_cmvn_stats_and_affine(kind, o, dtype):
    import torch
    from speech_amd import api
    lens = [16000, 400, 30000, 561, 8000]
    plan, b, Y = _run_batch(kind, lens, dtype, **o)
    y0 = Y.cpu().numpy()
    tol = 1e-10 if dtype == "f64" else 2e-5        # E[x^2] - mean^2 in another summation order: a few digits of the variance
    # corpus-wide statistics
    s, q, n = b.stats(Y)
    allf = np.concatenate(_utts(b, y0)).astype(np.float64)
    assert n == allf.shape[0]
    assert np.allclose(s, allf.sum(0), rtol=1e-12, atol=1e-9) and np.allclose(q, (allf * allf).sum(0), rtol=1e-12, atol=1e-9)
    # per-utterance CMVN, both modes, against the oracle
    for nv in (False, True):
        Z = Y.clone()
        b.cmvn(Z, norm_vars=nv)
        torch.cuda.synchronize()
        for got, y in zip(_utts(b, Z.cpu().numpy()), _utts(b, y0)):
            want = PO.cmvn(y, nv)
            assert np.abs(got - want).max() <= tol * max(1.0, np.abs(want).max()), (kind, nv)
    # corpus-wide CMVN through merged statistics and the affine pass
    scale, offset = api.cmvn_from_stats(*api.merge_stats(s, q, n))
    Z = Y.clone()
    b.affine(Z, scale, offset)
    torch.cuda.synchronize()
    z = np.concatenate(_utts(b, Z.cpu().numpy()))
    assert np.array_equal(z, PO.affine(np.concatenate(_utts(b, y0)), scale, offset))
    assert np.abs(z.mean(0)).max() < (1e-9 if dtype == "f64" else 1e-3) and np.abs(z.std(0) - 1).max() < (1e-9 if dtype == "f64" else 1e-3)


@pytest.mark.gpu
def test_cmvn_without_variance_is_mn0():
    """norm_vars = 0 is the reference's mn0 post-pass (intended D-wide form, c/kaldi_fbank.c:355-370)"""
    import torch
    lens = [16000, 4000]
    _, b, Y = _run_batch("fbank", lens, B=40)
    _, b2, Y2 = _run_batch("fbank", lens, B=40, mn0=1)
    b.cmvn(Y, norm_vars=False)
    torch.cuda.synchronize()
    assert np.abs(Y.cpu().numpy() - Y2.cpu().numpy()).max() < 2e-5


KINDS_D = [("mfcc", {}), ("fbank", dict(B=40)), ("fbank", dict(B=41, use_energy=1)), ("spectrogram", {})]      # D = 13, 40 (16-byte rows), 42, 257


@pytest.mark.gpu
@pytest.mark.parametrize("kind,o", KINDS_D)
@pytest.mark.parametrize("dtype", ["f32", "f64"])
def test_add_deltas_and_splice(kind, o, dtype):
    import torch
    lens = [16000, 400, 9000, 561]      # 561 samples: 2 frames, 400: 1 frame -- every tap clamps
    plan, b, Y = _run_batch(kind, lens, dtype, **o)
    y0 = Y.cpu().numpy()
    for order, window in ((2, 2), (1, 3), (3, 1), (0, 2)):
        Z = torch.zeros(b.total_out * (order + 1), device="cuda", dtype=Y.dtype)
        b.add_deltas(Y, Z, order, window)
        torch.cuda.synchronize()
        for got, y in zip(_utts(b, Z.cpu().numpy(), order + 1), _utts(b, y0)):
            want = PO.add_deltas(y, order, window)
            assert got.shape == want.shape
            assert np.abs(got - want).max() <= (1e-12 if dtype == "f64" else 1e-5) * max(1.0, np.abs(want).max()), (order, window)
            assert np.array_equal(got[:, :plan.D], y)
    for left, right in ((4, 4), (0, 2), (3, 0), (0, 0)):
        K = left + right + 1
        Z = torch.zeros(b.total_out * K, device="cuda", dtype=Y.dtype)
        b.splice(Y, Z, left, right)
        torch.cuda.synchronize()
        for got, y in zip(_utts(b, Z.cpu().numpy(), K), _utts(b, y0)):
            assert np.array_equal(got, PO.splice(y, left, right)), (left, right)


def test_oracle_subsample():
    y = np.arange(22, dtype=np.float32).reshape(11, 2)
    assert np.array_equal(PO.subsample(y, 3)[:, 0], [0, 6, 12, 18]) and np.array_equal(PO.subsample(y, 3, 2)[:, 0], [4, 10, 16])
    assert PO.subsample(y[:2], 3, 2).shape == (0, 2)


@pytest.mark.gpu
@pytest.mark.parametrize("kind,o", KINDS_D)
@pytest.mark.parametrize("dtype", ["f32", "f64"])
def test_splice_subsample(kind, o, dtype):
    """splice-feats | subsample-feats in one pass; utterances packed back to back; the (n, offset) table is re-made on change"""
    import torch
    lens = [16000, 400, 9000, 561, 720, 30000]      # 400 samples: 1 frame, 561: 2 frames, 720: 3 frames
    plan, b, Y = _run_batch(kind, lens, dtype, **o)
    y0 = Y.cpu().numpy()
    for left, right, n, off in ((0, 0, 3, 0), (1, 1, 3, 1), (3, 3, 3, 2), (0, 0, 1, 0), (2, 0, 2, 1), (0, 4, 7, 6), (1, 1, 3, 1)):
        K = left + right + 1
        rows, tot = b.subsample_rows(n, off)
        assert tot == rows.sum()
        Z = torch.full((max(1, tot * plan.D * K),), 777.0, device="cuda", dtype=Y.dtype)
        b.splice_subsample(Y, Z, left, right, n, off)
        torch.cuda.synchronize()
        z = Z.cpu().numpy()
        r0 = 0
        for u, y in enumerate(_utts(b, y0)):
            want = PO.subsample(PO.splice(y, left, right), n, off)
            assert want.shape[0] == rows[u], (u, n, off)
            got = z[r0 * plan.D * K:(r0 + rows[u]) * plan.D * K].reshape(-1, plan.D * K)
            assert np.array_equal(got, want), (left, right, n, off, u)
            r0 += int(rows[u])
        assert r0 == tot and (tot * plan.D * K == z.size or np.all(z[tot * plan.D * K:] == 777.0))
    # n = 1 without context is a copy in the dense packing
    Z = torch.zeros(b.total_out, device="cuda", dtype=Y.dtype)
    b.splice_subsample(Y, Z, 0, 0, 1, 0)
    torch.cuda.synchronize()
    assert np.array_equal(Z.cpu().numpy(), np.concatenate([y.ravel() for y in _utts(b, y0)]))


@pytest.mark.gpu
def test_post_argument_errors():
    import torch
    from speech_amd import api
    _, b, Y = _run_batch("mfcc", [4000])
    Z = torch.zeros(b.total_out * 3, device="cuda")
    for call in (lambda: b.add_deltas(Y, Z, 4, 2), lambda: b.add_deltas(Y, Z, 2, 0), lambda: b.add_deltas(Y, Y, 2, 2),
                 lambda: b.splice(Y, Z, -1, 0), lambda: b.splice(Y, Y, 1, 1), lambda: b.affine(Y, np.ones(3), np.zeros(3)),
                 lambda: b.splice_subsample(Y, Z, 0, 0, 0, 0), lambda: b.splice_subsample(Y, Z, 0, 0, 3, 3), lambda: b.splice_subsample(Y, Z, 0, 0, 3, -1),
                 lambda: b.subsample_rows(0), lambda: b.splice_subsample(Y, Y, 0, 0, 2, 0)):
        with pytest.raises((api.SpeechError, ValueError)):
            call()


@pytest.mark.gpu
def test_splice_with_unaligned_offsets_and_views():
    """rows of 40 floats are copied in 16-byte units only when every utterance's offset and both arrays allow it"""
    import torch
    from speech_amd import api
    lens = [8000, 561, 4000]
    plan = api.Plan("fbank", B=40)
    W = [plan.num_frames(n) for n in lens]
    X = torch.from_numpy(np.concatenate([signals.utterance(n, 500 + i) for i, n in enumerate(lens)])).cuda()
    for gap in (0, 1, 2):                                                            # gap 1: odd offsets; 2: offsets = 2 mod 4; 0: dense
        y_off = np.cumsum([0] + [w * 40 + gap for w in W[:-1]]).astype(np.uint64)
        b = api.Batch(plan, lens, y_off=y_off)
        nel = int(y_off[-1]) + W[-1] * 40
        Ybuf = torch.zeros(nel + 8, device="cuda")
        for shift in (0, 1):                                                         # 1: arrays that start 4 bytes off a 16-byte boundary
            Y = Ybuf[shift:shift + nel]
            b.run(X, Y)
            Z = torch.zeros(nel * 3 + 4, device="cuda")[shift:]
            b.splice(Y, Z, 1, 1)
            torch.cuda.synchronize()
            y0, z = Y.cpu().numpy(), Z.cpu().numpy()
            for u in range(3):
                y = y0[int(y_off[u]):int(y_off[u]) + W[u] * 40].reshape(-1, 40)
                got = z[int(y_off[u]) * 3:int(y_off[u]) * 3 + W[u] * 120].reshape(-1, 120)
                assert np.array_equal(got, PO.splice(y, 1, 1)), (gap, shift, u)


@pytest.mark.gpu
@pytest.mark.parametrize("nutt", [64, 65, 4097, 9000])
def test_stats_tree_over_many_utterances(nutt):
    """the corpus sums are reduced on the device in groups of 64 utterances (1, 2 and 3 levels here) and repeat exactly"""
    import torch
    from speech_amd import api
    plan = api.Plan("mfcc", snip_edges=0)
    lens = [160 * (1 + (7 * i) % 5) for i in range(nutt)]
    b = api.Batch(plan, lens)
    Y = (torch.randn(b.total_out, device="cuda") * 3 + 1)
    s, q, n = b.stats(Y)
    y = Y.cpu().numpy().astype(np.float64).reshape(-1, plan.D)
    assert n == y.shape[0] == b.total_frames
    assert np.allclose(s, y.sum(0), rtol=1e-11, atol=1e-8) and np.allclose(q, (y * y).sum(0), rtol=1e-11, atol=1e-8)
    s2, q2, _ = b.stats(Y)
    assert np.array_equal(s, s2) and np.array_equal(q, q2)
    # the affine slot and the tree share the batch's buffer with the per-utterance moments: one after the other still works
    Z = Y.clone()
    b.affine(Z, np.full(plan.D, 2.0), np.full(plan.D, -1.0))
    b.cmvn(Z, norm_vars=True)
    s3, q3, _ = b.stats(Z)
    multi = sum(int(w) for w in b.n_frames if w > 1)          # a one-frame utterance normalises to zeros
    assert np.abs(s3).max() < 1e-2 * nutt and np.allclose(q3 / multi, 1.0, atol=1e-3)
