"""Corpus-scale extraction in HBM-sized waves, with a manifest and resume at wave granularity (SURVEY.md section 5
"checkpoint / resume", 8(d) configs 3 and 4).

The reference processes one utterance per process (src/kaldi_mfcc.cpp:283-308) and has no notion of a corpus.  Here a
rank's share of a corpus (api.shard_utterances: no data-path collective) is cut into WAVES -- consecutive runs of its
utterances whose audio fits a budget of HBM -- and every wave is one resident batch: audio in (generated on the device
for the synthetic corpus, or read from wav files), one launch of the per-frame pipeline, features out.  A wave that has
been written is recorded as one line of the rank's manifest; a run that is killed and started again skips the recorded
waves (after checking that the manifest belongs to the same corpus, options and sharding).

  SyntheticCorpus      the integer-only corpus of SURVEY.md 8(d): lengths and samples as functions of (seed, utterance)
  synth_utterance      numpy statement of speech_internal.hpp's synth_sample (bit-identical to speech_batch_synth_audio)
  plan_waves / walk    host-side bookkeeping (pure Python: tested on CPU, including kill-and-resume)
  run_synthetic        the device loop for the synthetic corpus (bench.py's corpus leg, tests)
"""
import hashlib
import json
import os
import time

import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _mix64(z):
    """speech_internal.hpp mix64 on numpy uint64 arrays (wrapping arithmetic)."""
    z = np.asarray(z, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = z + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return z ^ (z >> np.uint64(31))


def synth_utterance(seed, utt, n, dtype=np.int16):
    """Samples 0..n-1 of corpus utterance `utt` under master seed `seed`: the lines of speech_internal.hpp (synth_utt,
    synth_sample) restated with numpy integers."""
    ku = _mix64(np.uint64(seed) ^ np.uint64(utt))
    idx = np.arange(n, dtype=np.uint64)
    with np.errstate(over="ignore"):
        h = _mix64(ku ^ (idx * np.uint64(0xD6E8FEB86659FD93)))
    f = np.uint64(0x3FFF)
    v = ((h & f) + ((h >> np.uint64(14)) & f) + ((h >> np.uint64(28)) & f) + ((h >> np.uint64(42)) & f)).astype(np.int64) - 32766
    v >>= 3                                                     # arithmetic shift: floor
    n32 = idx.astype(np.uint32)
    for k in range(3):
        with np.errstate(over="ignore"):
            r = _mix64(ku + np.uint64(k + 1))
        ph = np.uint32(int(r) & 0xFFFFFFFF)
        inc = np.uint32(0x0147AE14 + (((int(r) >> 32) * 0x71EB851F) >> 32))
        amp = 500 + ((int(r) >> 20) & 2047)
        with np.errstate(over="ignore"):
            t = ((ph + n32 * inc) >> np.uint32(16)).astype(np.int64)
        tri = np.where(t < 32768, t, 65535 - t)
        v += ((tri - 16384) * amp) >> 14
    return v.astype(dtype)


class SyntheticCorpus:
    """`hours` of 16 kHz utterances of 1-30 s (uniform in samples), utterance u's length and samples functions of (seed, u)."""

    def __init__(self, hours=None, utterances=None, seed=0x5EED2026, sr=16000, min_samples=16000, max_samples=480000):
        self.seed, self.sr, self.min_samples, self.max_samples = int(seed), int(sr), int(min_samples), int(max_samples)
        span = self.max_samples - self.min_samples + 1
        if utterances is None:
            target = int(round(float(hours) * 3600.0 * self.sr))
            lens = np.zeros(0, dtype=np.int64)
            while lens.sum() < target:                          # grow the list until it covers the target, then cut
                u = np.arange(lens.size, lens.size + max(1024, int(1.1 * (target - lens.sum()) / (0.5 * (self.min_samples + self.max_samples))) + 1), dtype=np.uint64)
                lens = np.concatenate([lens, self.min_samples + (_mix64(np.uint64(self.seed ^ 0xC0FFEE) ^ u) % np.uint64(span)).astype(np.int64)])
            utterances = int(np.searchsorted(np.cumsum(lens), target) + 1)
            self.lens = lens[:utterances]
        else:
            u = np.arange(int(utterances), dtype=np.uint64)
            self.lens = self.min_samples + (_mix64(np.uint64(self.seed ^ 0xC0FFEE) ^ u) % np.uint64(span)).astype(np.int64)
        self.hours = float(self.lens.sum()) / self.sr / 3600.0

    def describe(self):
        return {"kind": "synthetic", "seed": self.seed, "sr": self.sr, "utterances": int(self.lens.size), "samples": int(self.lens.sum()),
                "min_samples": self.min_samples, "max_samples": self.max_samples}

    def audio(self, utt, dtype=np.int16):
        return synth_utterance(self.seed, utt, int(self.lens[utt]), dtype)


def plan_waves(ids, lens, wave_samples):
    """Cut a rank's utterances (corpus numbers `ids`, in order; their lengths `lens`) into waves of at most `wave_samples`
    samples (at least one utterance each).  Returns [(first, count)] as positions in `ids`."""
    waves, first, acc = [], 0, 0
    for i, n in enumerate(lens):
        n = int(n)
        if i > first and acc + n > wave_samples:
            waves.append((first, i - first))
            first, acc = i, 0
        acc += n
    if len(lens) > first:
        waves.append((first, len(lens) - first))
    return waves


def _signature(corpus_desc, kind, opts, rank, world, wave_samples):
    blob = json.dumps({"corpus": corpus_desc, "kind": kind, "opts": {k: opts[k] for k in sorted(opts)}, "rank": rank, "world": world,
                       "wave_samples": int(wave_samples)}, sort_keys=True)
    return hashlib.sha256(blob.encode()).hexdigest()


class Manifest:
    """One JSON line per finished wave, after a header line that names what the manifest belongs to.  A line is appended
    (and flushed to disk) only after the wave's output exists, so whatever the file lists is complete."""

    def __init__(self, path, signature, header=None):
        self.path, self.signature = path, signature
        self.done = {}
        if path is None:
            return
        if os.path.exists(path) and os.path.getsize(path) > 0:
            with open(path) as f:
                lines = [ln for ln in f.read().split("\n") if ln.strip()]
            head = json.loads(lines[0])
            if head.get("signature") != signature:
                raise ValueError("%s belongs to another corpus / option set / sharding (signature %s, this run %s): "
                                 "remove it or choose another output directory" % (path, head.get("signature"), signature))
            good = [lines[0]]
            for ln in lines[1:]:
                try:
                    rec = json.loads(ln)
                    self.done[int(rec["wave"])] = rec
                    good.append(ln)
                except (ValueError, KeyError, TypeError):
                    pass                                        # a line cut short by the kill: that wave is redone
            if len(good) != len(lines) or not open(path).read().endswith("\n"):
                tmp = path + ".tmp"                             # drop the broken line, so that new lines start on a line of their own
                with open(tmp, "w") as f:
                    f.write("\n".join(good) + "\n")
                    f.flush(); os.fsync(f.fileno())
                os.replace(tmp, path)
        else:
            os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
            with open(path, "w") as f:
                f.write(json.dumps(dict(header or {}, signature=signature)) + "\n")
                f.flush(); os.fsync(f.fileno())

    def record(self, wave, rec):
        rec = dict(rec, wave=int(wave))
        self.done[int(wave)] = rec
        if self.path is not None:
            with open(self.path, "a") as f:
                f.write(json.dumps(rec) + "\n")
                f.flush(); os.fsync(f.fileno())


def walk(waves, manifest, do_wave, output_exists=None):
    """Run do_wave(w, first, count) -> record for every wave the manifest does not list (or whose output is gone), in order.
    Returns (waves run now, waves skipped)."""
    ran = skipped = 0
    for w, (first, count) in enumerate(waves):
        rec = manifest.done.get(w)
        if rec is not None and rec.get("first") == first and rec.get("count") == count and (output_exists is None or output_exists(rec)):
            skipped += 1
            continue
        rec = do_wave(w, first, count)
        manifest.record(w, dict(rec, first=first, count=count))
        ran += 1
    return ran, skipped


def run_synthetic(spec, kind, opts, device=0, wave_audio_s=1.0e6, out_dir=None, rank=0, world=1, keep_output=True, pcm16=True, max_waves=None):
    """Extract this rank's share of a SyntheticCorpus in waves of at most `wave_audio_s` audio-seconds.  Each wave: build
    the batch, generate its audio on the device (speech_batch_synth_audio), ONE launch of the per-frame pipeline, then --
    with out_dir -- copy the features back and write them as a Kaldi archive `feats.<rank>.<wave>.ark`, recorded in
    `manifest.<rank>.jsonl`; without out_dir only a checksum of the features is kept (bench).  Resumes from the manifest.
    Returns totals; kernel_ms / generate_ms are HIP-event times of the extraction launches / the generator."""
    import torch
    from . import api, ark
    lens_all = spec.lens
    mine = api.shard_utterances(lens_all, world)[rank] if world > 1 else np.arange(lens_all.size, dtype=np.int64)
    lens = lens_all[mine]
    wave_samples = int(wave_audio_s * spec.sr)
    waves = plan_waves(mine, lens, wave_samples)
    if max_waves is not None:
        waves = waves[:max_waves]
    sig = _signature(spec.describe(), kind, opts, rank, world, wave_samples)
    mpath = os.path.join(out_dir, "manifest.%d.jsonl" % rank) if out_dir else None
    man = Manifest(mpath, sig, header={"corpus": spec.describe(), "kind": kind, "rank": rank, "world": world})
    plan = api.Plan(kind, device=device, **opts)
    dev = torch.device("cuda", device)
    cap_x = max(int(lens[f:f + c].sum()) for f, c in waves) if waves else 0
    cap_y = max(int(sum(plan.num_frames(int(n)) for n in lens[f:f + c])) for f, c in waves) * plan.D if waves else 0
    X = torch.empty(cap_x, device=dev, dtype=torch.int16 if pcm16 else torch.float32)
    Y = torch.empty(cap_y, device=dev, dtype=torch.float32)
    stream = torch.cuda.current_stream().cuda_stream
    tot = {"kernel_ms": 0.0, "generate_ms": 0.0, "samples": 0, "features": 0, "frames": 0, "utterances": 0}
    t0 = time.perf_counter()

    def do_wave(w, first, count):
        b = api.Batch(plan, lens[first:first + count])
        e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        e[0].record()
        b.synth_audio(X, seed=spec.seed, ids=mine[first:first + count], stream=stream)
        e[1].record()
        b.run(X, Y, stream=stream)
        e[2].record()
        e[2].synchronize()
        rec = {"utterances": count, "samples": b.total_samples, "frames": b.total_frames, "features": b.total_out,
               "generate_ms": e[0].elapsed_time(e[1]), "kernel_ms": e[1].elapsed_time(e[2])}
        if out_dir and keep_output:
            y = Y[:b.total_out].cpu().numpy()
            name = "feats.%d.%d.ark" % (rank, w)
            tmp = os.path.join(out_dir, name + ".tmp")
            ark.write_ark(tmp, [("utt%09d" % int(mine[first + u]), y[int(b.y_off[u]):int(b.y_off[u]) + int(b.n_frames[u]) * plan.D].reshape(-1, plan.D))
                                for u in range(count)])
            os.replace(tmp, os.path.join(out_dir, name))        # the archive appears under its name only when complete
            rec["ark"] = name
            rec["bytes"] = os.path.getsize(os.path.join(out_dir, name))
        else:
            rec["checksum"] = int(Y[:b.total_out].view(torch.int32).sum(dtype=torch.int64).item())
        b.close()
        for k in ("kernel_ms", "generate_ms", "samples", "features", "frames", "utterances"):
            tot[k] += rec[k]
        return rec

    def exists(rec):
        return "ark" not in rec or (os.path.exists(os.path.join(out_dir, rec["ark"])) and os.path.getsize(os.path.join(out_dir, rec["ark"])) == rec.get("bytes"))

    ran, skipped = walk(waves, man, do_wave, exists if out_dir else None)
    torch.cuda.synchronize()
    plan.close()
    # Totals are whole-corpus figures, built from the manifest's records like run_files does (a resumed run skips finished
    # waves: their samples, frames and kernel times come from the lines the earlier run wrote); what THIS call ran is kept
    # beside them as *_run_now, so that a rate computed after a resume says which part it covers.
    recs = [man.done[w] for w in sorted(man.done) if w < len(waves)]
    run_now = dict(tot)
    for k in ("kernel_ms", "generate_ms", "samples", "features", "frames", "utterances"):
        tot[k] = type(run_now[k])(sum(r.get(k, 0) for r in recs))
        tot[k + "_run_now"] = run_now[k]
    tot.update(waves=len(waves), waves_run=ran, waves_skipped=skipped, wall_s=time.perf_counter() - t0, audio_s=tot["samples"] / float(spec.sr),
               audio_s_run_now=run_now["samples"] / float(spec.sr), manifest=mpath, records=recs)
    return tot


def apply_cmvn_to_archives(arks, scale, offset, _after_each=None):
    """Write every archive of `arks` (feats.<rank>.<wave>.ark) normalised, y = x * scale + offset, as feats_cmvn.<rank>.<wave>.ark
    + .scp BESIDE it.  The extracted archives are never modified, so the step can be killed and run again any number of times:
    a normalised archive is always computed from the raw one (never from an already normalised one), written under a
    temporary name and renamed.  Returns the normalised archives' paths.  _after_each(i): test hook, called after archive i."""
    from . import ark
    out = []
    for i, path in enumerate(arks):
        d, base = os.path.dirname(path), os.path.basename(path)
        if not base.startswith("feats."):
            raise ValueError("unexpected archive name %s" % path)
        dst = os.path.join(d, "feats_cmvn." + base[len("feats."):])
        feats = [(k, (m * scale.astype(m.dtype)).astype(m.dtype) + offset.astype(m.dtype)) for k, _, m in ark.read_ark(path)]
        scp = dst[:-4] + ".scp"
        open(scp + ".tmp", "w").close()
        ark.write_ark(dst + ".tmp", feats, scp=scp + ".tmp")
        with open(scp + ".tmp") as f:
            lines = f.read().replace(dst + ".tmp", dst)
        with open(scp + ".tmp", "w") as f:
            f.write(lines)
        os.replace(scp + ".tmp", scp)
        os.replace(dst + ".tmp", dst)
        out.append(dst)
        if _after_each is not None:
            _after_each(i)
    return out


def read_wav16(path):
    """One channel of a 16-bit PCM .wav as int16, and its sample rate."""
    import wave
    with wave.open(path, "rb") as w:
        if w.getsampwidth() != 2:
            raise ValueError("%s: only 16-bit PCM" % path)
        x = np.frombuffer(w.readframes(w.getnframes()), dtype=np.int16)
        return x[::w.getnchannels()].copy(), w.getframerate()


def run_files(keys, paths, kind, opts, out_dir, rank=0, world=1, wave_bytes=1 << 30, device=0, extract=None, reader=read_wav16):
    """Extract this rank's share of a list of .wav files in waves of at most `wave_bytes` of file size, so that host memory
    holds one wave of audio and features at a time, never the shard (examples/extract_corpus.py).  A wave: read its files,
    one call of speech_plan_run_host (int16 over PCIe, dither keyed by the utterance's line number), write
    `feats.<rank>.<wave>.ark` + `.scp` under a temporary name, rename, record the wave -- with its per-feature sums, so a
    resumed run needs no second look at finished waves -- in `manifest.<rank>.jsonl`.  `extract(xs, ids, rate) -> [W x D]`
    replaces the device call (tests).  Returns {"sum", "sumsq", "frames", "waves", "waves_run", "waves_skipped", "arks"}."""
    from . import api, ark
    sizes = np.array([os.path.getsize(p) for p in paths], dtype=np.int64)
    mine = api.shard_utterances(sizes, world)[rank] if world > 1 else np.arange(sizes.size, dtype=np.int64)
    waves = plan_waves(mine, sizes[mine], int(wave_bytes))
    desc = {"kind": "files", "n": int(sizes.size), "bytes": int(sizes.sum()), "names": hashlib.sha256("\n".join(keys).encode()).hexdigest()}
    sig = _signature(desc, kind, opts, rank, world, int(wave_bytes))
    os.makedirs(out_dir, exist_ok=True)
    man = Manifest(os.path.join(out_dir, "manifest.%d.jsonl" % rank), sig, header={"corpus": desc, "kind": kind, "rank": rank, "world": world})
    state = {"plan": None, "rate": None}

    def device_extract(xs, ids, rate):
        if state["plan"] is None:
            o = dict(opts)
            o.setdefault("sr", float(rate))
            o.setdefault("hif", 0.5 * float(rate))
            state["plan"] = api.Plan(kind, device=device, **o)
        return state["plan"].run_host(xs, ids=ids)

    def do_wave(w, first, count):
        ids = mine[first:first + count]
        xs = []
        for i in ids:
            x, r = reader(paths[int(i)])
            state["rate"] = state["rate"] or r
            if r != state["rate"]:
                raise ValueError("%s: sample rate %d differs from the corpus's %d" % (paths[int(i)], r, state["rate"]))
            xs.append(x)
        feats = (extract or device_extract)(xs, ids, state["rate"])
        name = "feats.%d.%d" % (rank, w)
        tmp_ark, tmp_scp = os.path.join(out_dir, name + ".ark.tmp"), os.path.join(out_dir, name + ".scp.tmp")
        open(tmp_scp, "w").close()
        ark.write_ark(tmp_ark, [(keys[int(i)], f) for i, f in zip(ids, feats)], scp=tmp_scp)
        with open(tmp_scp) as f:                                # the script file names the archive's final path
            lines = f.read().replace(tmp_ark, os.path.join(out_dir, name + ".ark"))
        with open(tmp_scp, "w") as f:
            f.write(lines)
        os.replace(tmp_scp, os.path.join(out_dir, name + ".scp"))
        os.replace(tmp_ark, os.path.join(out_dir, name + ".ark"))    # last: what the manifest's existence check looks at
        D = feats[0].shape[1] if feats else 0
        s, q = np.zeros(D), np.zeros(D)
        for f in feats:
            f64 = f.astype(np.float64)
            s += f64.sum(0)
            q += (f64 * f64).sum(0)
        return {"ark": name + ".ark", "bytes": os.path.getsize(os.path.join(out_dir, name + ".ark")), "utterances": int(count),
                "frames": int(sum(f.shape[0] for f in feats)), "sum": s.tolist(), "sumsq": q.tolist()}

    def exists(rec):
        p = os.path.join(out_dir, rec["ark"])
        return os.path.exists(p) and os.path.getsize(p) == rec.get("bytes")

    ran, skipped = walk(waves, man, do_wave, exists)
    if state["plan"] is not None:
        state["plan"].close()
    recs = [man.done[w] for w in sorted(man.done) if w < len(waves)]
    D = max([len(r["sum"]) for r in recs] + [0])
    s, q = np.zeros(D), np.zeros(D)
    for r in recs:
        if len(r["sum"]) == D:
            s += np.array(r["sum"]); q += np.array(r["sumsq"])
    return {"sum": s, "sumsq": q, "frames": int(sum(r["frames"] for r in recs)), "utterances": int(sum(r["utterances"] for r in recs)),
            "waves": len(waves), "waves_run": ran, "waves_skipped": skipped, "arks": [os.path.join(out_dir, r["ark"]) for r in recs]}
