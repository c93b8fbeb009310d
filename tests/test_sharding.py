"""Multi-GPU path on CPU: utterances are sharded over ranks with no collective on the data path.
Runs two real processes over gloo; each rank derives its own shard, and the only exchange is the bookkeeping the
bench does (a max over ranks)."""
import os
import socket

import numpy as np
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
