"""Corpus extraction over the GPUs of one node, one process per GPU, no data-path collective.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node=8 --master-addr 127.0.0.1 \\
        examples/extract_corpus.py wav.scp out_dir --kind mfcc [--cmvn] [--opt B=40 --opt C=40 ...]

wav.scp holds "<utt-id> <path to 16-bit PCM .wav>" lines.  Every rank takes its share of the utterances (longest first,
least-loaded rank: api.shard_utterances on the file sizes), reads them, extracts them with speech_plan_run_host (int16 over
PCIe, dither keyed by the utterance's line number, so the features do not depend on the number of ranks) and writes
out_dir/feats.<rank>.ark + .scp.  With --cmvn the ranks add their per-feature sums with ONE all-reduce of 2 D + 1 doubles
(api.merge_stats: RCCL, or gloo when the ranks share a GPU) and write corpus-normalised features.  Without a launcher it
runs as a single rank."""
import argparse
import os
import sys
import wave

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from speech_amd import api, ark  # noqa: E402


def read_wav(path):
    with wave.open(path, "rb") as w:
        if w.getsampwidth() != 2:
            raise ValueError("%s: only 16-bit PCM" % path)
        x = np.frombuffer(w.readframes(w.getnframes()), dtype=np.int16)
        return x[::w.getnchannels()].copy(), w.getframerate()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("wav_scp")
    ap.add_argument("out_dir")
    ap.add_argument("--kind", default="mfcc", choices=["spectrogram", "fbank", "mfcc", "plp"])
    ap.add_argument("--cmvn", action="store_true", help="corpus-wide mean / variance normalisation")
    ap.add_argument("--opt", action="append", default=[], help="plan option, name=value (e.g. B=40, dither=1.0, snip_edges=0)")
    a = ap.parse_args()

    rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    ndev = api._lib.load().speech_device_count()
    if ndev < 1:
        raise SystemExit("no HIP device")
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl" if world <= ndev else "gloo")          # ranks that share a GPU cannot form an RCCL group
        if world <= ndev:
            import torch
            torch.cuda.set_device(local)

    entries = [ln.split(None, 1) for ln in open(a.wav_scp) if ln.strip()]
    keys, paths = [e[0] for e in entries], [e[1].strip() for e in entries]
    sizes = [os.path.getsize(p) for p in paths]
    mine = api.shard_utterances(sizes, world)[rank]
    opts = {}
    for o in a.opt:
        k, v = o.split("=", 1)
        opts[k] = v if k in ("win_type", "win") else (float(v) if "." in v or "e" in v.lower() else int(v))
    xs, rate = [], None
    for i in mine:
        x, r = read_wav(paths[int(i)])
        rate = rate or r
        if r != rate:
            raise SystemExit("sample rates differ")
        xs.append(x)
    if rate is not None:
        opts.setdefault("sr", float(rate))
        opts.setdefault("hif", 0.5 * float(rate))
    plan = api.Plan(a.kind, device=local % ndev, **opts) if xs or world == 1 else None
    feats = plan.run_host(xs, ids=mine) if xs else []

    if a.cmvn:
        D = plan.D if plan is not None else 0
        if world > 1:                                                          # a rank without utterances still joins the all-reduce
            import torch
            import torch.distributed as dist
            d = torch.tensor([D], dtype=torch.int64)
            if dist.get_backend() == "nccl":
                d = d.cuda()
            dist.all_reduce(d, op=dist.ReduceOp.MAX)
            D = int(d.item())
        s, q = np.zeros(D), np.zeros(D)
        for f in feats:
            f64 = f.astype(np.float64)
            s += f64.sum(0)
            q += (f64 * f64).sum(0)
        s, q, n = api.merge_stats(s, q, sum(f.shape[0] for f in feats))
        scale, offset = api.cmvn_from_stats(s, q, n)
        feats = [(f * scale.astype(f.dtype)).astype(f.dtype) + offset.astype(f.dtype) for f in feats]

    os.makedirs(a.out_dir, exist_ok=True)
    out = os.path.join(a.out_dir, "feats.%d.ark" % rank)
    scp = os.path.join(a.out_dir, "feats.%d.scp" % rank)
    open(scp, "w").close()
    ark.write_ark(out, [(keys[int(i)], f) for i, f in zip(mine, feats)], scp=scp)
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()
    print("rank %d of %d: %d utterances, %d frames -> %s" % (rank, world, len(feats), sum(f.shape[0] for f in feats), out), flush=True)


if __name__ == "__main__":
    main()
