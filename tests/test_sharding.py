"""Multi-GPU path on CPU: utterances are sharded over ranks with no collective on the data path.
Runs two real processes over gloo; each rank derives its own shard, and the only exchange is the bookkeeping the
bench does (a max over ranks)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from speech_amd import api


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, lens, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    parts = api.shard_utterances(lens, world)
    mine = parts[rank]
    # per-rank geometry exactly as a rank would build it for its own utterances
    frames = np.array([api.frame_sizes(int(n))[0] for n in lens[mine]], dtype=np.int64)
    t = torch.tensor([float(lens[mine].sum())], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)          # the only cross-rank step: a max, as in bench.py
    cover = torch.zeros(len(lens), dtype=torch.int64)
    cover[torch.from_numpy(mine)] = 1
    dist.all_reduce(cover)                            # test bookkeeping only
    q.put((rank, mine.tolist(), int(frames.sum()), float(t.item()), cover.tolist()))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_cover_every_utterance_once():
    rng = np.random.default_rng(11)
    lens = rng.integers(16000, 480001, size=257).astype(np.int64)
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, lens, q)) for r in range(2)]
    for p in procs:
        p.start()
    outs = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    a, b = set(outs[0][1]), set(outs[1][1])
    assert not (a & b) and (a | b) == set(range(len(lens)))
    assert all(c == 1 for c in outs[0][4])
    loads = [int(lens[list(s)].sum()) for s in (a, b)]
    assert abs(loads[0] - loads[1]) <= int(lens.max())           # LPT: imbalance below one utterance
    assert outs[0][3] == outs[1][3] == float(max(loads))
    total_frames = sum(api.frame_sizes(int(n))[0] for n in lens)
    assert outs[0][2] + outs[1][2] == total_frames


def test_shard_is_deterministic_and_never_splits_an_utterance():
    lens = np.array([480000, 16000, 16000, 32000, 240000, 100000], dtype=np.int64)
    p1 = api.shard_utterances(lens, 4)
    p2 = api.shard_utterances(lens, 4)
    assert all(np.array_equal(x, y) for x, y in zip(p1, p2))
    assert sorted(np.concatenate(p1).tolist()) == list(range(6))
    assert api.shard_utterances(lens, 1)[0].tolist() == list(range(6))


@pytest.mark.gpu
def test_corpus_extraction_two_ranks_on_the_gpu(tmp_path):
    """examples/extract_corpus.py as two processes (torch.distributed.run, both on GPU 0, gloo for the one all-reduce):
    the ranks' archives together hold the single-process features bit for bit -- dither on, so the noise follows the
    utterance, not the rank -- and the corpus-wide CMVN of both runs agrees"""
    import subprocess
    import sys
    import wave
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
    import signals
    from speech_amd import ark
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    lines = []
    for i, n in enumerate([16000, 4000, 30000, 8000, 12000, 2000, 20000]):
        p = tmp_path / ("u%d.wav" % i)
        with wave.open(str(p), "wb") as w:
            w.setnchannels(1); w.setsampwidth(2); w.setframerate(16000)
            w.writeframes(signals.utterance(n, 700 + i).astype(np.int16).tobytes())
        lines.append("utt%d %s" % (i, p))
    lst = tmp_path / "wav.scp"
    lst.write_text("\n".join(lines) + "\n")
    script = os.path.join(root, "examples", "extract_corpus.py")
    # --wave-mb 0: every utterance a wave of its own -- the streaming path with as many waves as it can have
    common = [str(lst), "--kind", "mfcc", "--cmvn", "--apply-cmvn", "--wave-mb", "0", "--opt", "dither=1.0", "--opt", "snip_edges=0"]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29541")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    r1 = subprocess.run([sys.executable, script, common[0], str(tmp_path / "one")] + common[1:], capture_output=True, env=env, timeout=300)
    assert r1.returncode == 0, r1.stderr.decode()[-2000:]
    r2 = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                         "--master-port", "29541", script, common[0], str(tmp_path / "two")] + common[1:], capture_output=True, env=env, timeout=300)
    assert r2.returncode == 0, r2.stderr.decode()[-2000:]
    import glob
    one = {}
    for path in sorted(glob.glob(str(tmp_path / "one" / "feats_cmvn.0.*.ark"))):      # --apply-cmvn: normalised copies beside the extracted archives
        one.update({k: m for k, _, m in ark.read_ark(path)})
    assert len(glob.glob(str(tmp_path / "one" / "feats.0.*.ark"))) == 7       # seven waves of one utterance
    two = {}
    for rank in (0, 1):
        part = {}
        for scp in sorted(glob.glob(str(tmp_path / "two" / ("feats_cmvn.%d.*.scp" % rank)))):
            part.update({k: m for k, m in ark.read_scp(scp)})
        assert part and not (set(part) & set(two))
        two.update(part)
    assert os.path.exists(tmp_path / "two" / "cmvn.npz") and os.path.exists(tmp_path / "two" / "manifest.1.jsonl")
    assert set(one) == set(two) == {"utt%d" % i for i in range(7)}
    for k in one:
        # the sums are added in another order over two ranks: scale / offset agree to rounding, not to the bit
        assert one[k].shape == two[k].shape and np.abs(one[k] - two[k]).max() < 1e-5, k
    allf = np.concatenate(list(two.values()))
    assert np.abs(allf.mean(0)).max() < 1e-3 and np.abs(allf.std(0) - 1).max() < 1e-3
