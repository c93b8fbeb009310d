"""Corpus extraction over the GPUs of one node, one process per GPU, no data-path collective.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node=8 --master-addr 127.0.0.1 \\
        examples/extract_corpus.py wav.scp out_dir --kind mfcc [--cmvn] [--wave-mb 1024] [--opt B=40 --opt C=40 ...]

wav.scp holds "<utt-id> <path to 16-bit PCM .wav>" lines.  Every rank takes its share of the utterances (longest first,
least-loaded rank: api.shard_utterances on the file sizes) and STREAMS it in waves of at most --wave-mb of audio
(speech_amd.corpus.run_files): read a wave's files, extract them with speech_plan_run_host (int16 over PCIe, dither keyed by
the utterance's line number, so the features do not depend on the number of ranks or on the wave size), write
out_dir/feats.<rank>.<wave>.ark + .scp, record the wave in out_dir/manifest.<rank>.jsonl.  Host memory holds one wave, not
the shard; a run that was killed resumes after its last recorded wave.  With --cmvn the ranks add their per-feature sums
(kept per wave in the manifest) with ONE all-reduce of 2 D + 1 doubles (api.merge_stats: RCCL, or gloo when the ranks share
a GPU) and write out_dir/cmvn.npz (scale, offset: y = x * scale + offset, what Kaldi's apply-cmvn does from its stats
file); --apply-cmvn then writes normalised copies feats_cmvn.<rank>.<wave>.ark beside the archives, again one wave at a time
(the extracted archives are never modified, so this step, too, can be killed and repeated).  Without a launcher it runs as a single
rank."""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from speech_amd import api, ark, corpus  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("wav_scp")
    ap.add_argument("out_dir")
    ap.add_argument("--kind", default="mfcc", choices=["spectrogram", "fbank", "mfcc", "plp"])
    ap.add_argument("--cmvn", action="store_true", help="corpus-wide mean / variance normalisation")
    ap.add_argument("--apply-cmvn", action="store_true", help="with --cmvn: write normalised copies feats_cmvn.* beside the archives")
    ap.add_argument("--wave-mb", type=int, default=1024, help="audio read, extracted and written at a time, MiB of .wav files")
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
    opts = {}
    for o in a.opt:
        k, v = o.split("=", 1)
        opts[k] = v if k in ("win_type", "win") else (float(v) if "." in v or "e" in v.lower() else int(v))
    res = corpus.run_files(keys, paths, a.kind, opts, a.out_dir, rank=rank, world=world, wave_bytes=a.wave_mb << 20, device=local % ndev)

    if a.cmvn:
        D = res["sum"].size
        if world > 1:                                                          # a rank without utterances still joins the all-reduce
            import torch
            import torch.distributed as dist
            d = torch.tensor([D], dtype=torch.int64)
            if dist.get_backend() == "nccl":
                d = d.cuda()
            dist.all_reduce(d, op=dist.ReduceOp.MAX)
            D = int(d.item())
        s = res["sum"] if res["sum"].size == D else np.zeros(D)
        q = res["sumsq"] if res["sumsq"].size == D else np.zeros(D)
        s, q, n = api.merge_stats(s, q, res["frames"])
        scale, offset = api.cmvn_from_stats(s, q, n)
        if rank == 0:
            np.savez(os.path.join(a.out_dir, "cmvn.npz"), scale=scale, offset=offset, frames=n)
        if a.apply_cmvn:
            # normalised copies beside the extracted archives (feats_cmvn.<rank>.<wave>.ark + .scp), one wave at a time; the
            # extracted archives stay as they are, so a run killed in this loop and started again normalises nothing twice
            corpus.apply_cmvn_to_archives(res["arks"], scale, offset)

    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()
    print("rank %d of %d: %d utterances, %d frames in %d waves (%d run now, %d already done) -> %s" %
          (rank, world, res["utterances"], res["frames"], res["waves"], res["waves_run"], res["waves_skipped"], a.out_dir), flush=True)


if __name__ == "__main__":
    main()
