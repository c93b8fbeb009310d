"""The corpus walker (speech_amd/corpus.py): synthetic corpus generator, waves, manifest and resume.
CPU tests cover the generator (numpy against the library's host statement of the same integer arithmetic) and the
bookkeeping, including a run that is killed between waves and started again; GPU tests cover the device generator and the
wave loop end to end against the oracle."""
import ctypes as C
import json
import os

import numpy as np
import pytest

from speech_amd import _lib, corpus


def test_numpy_generator_equals_the_library_host_generator():
    lib = _lib.load()
    for seed, utt, n in [(0x5EED2026, 0, 4099), (0x5EED2026, 99999, 16000), (1, 2 ** 40 + 3, 70001), (0xFFFFFFFFFFFFFFFF, 17, 1)]:
        out = np.zeros(n, dtype=np.int16)
        assert lib.speech_synth_audio_host(seed, utt, n, out.ctypes.data_as(C.POINTER(C.c_short))) == 0
        assert np.array_equal(out, corpus.synth_utterance(seed, utt, n)), (seed, utt)
    x = corpus.synth_utterance(0x5EED2026, 5, 160000).astype(np.float64)
    assert np.abs(x).max() < 12000 and 500.0 < x.std() < 4000.0            # int16 scale, around -25 dBFS
    assert not np.array_equal(corpus.synth_utterance(0x5EED2026, 5, 100), corpus.synth_utterance(0x5EED2026, 6, 100))


def test_synthetic_corpus_lengths():
    c = corpus.SyntheticCorpus(hours=2.0)
    assert abs(c.hours - 2.0) < 30.0 / 3600.0 and c.lens.min() >= 16000 and c.lens.max() <= 480000
    # utterance u's length depends on (seed, u) only: a longer corpus starts with the same utterances
    c2 = corpus.SyntheticCorpus(hours=3.0)
    assert np.array_equal(c2.lens[:c.lens.size], c.lens)
    assert np.array_equal(corpus.SyntheticCorpus(utterances=50).lens, c.lens[:50])
    assert abs(c.lens.mean() / 16000.0 - 15.5) < 1.5


def test_plan_waves_covers_every_utterance_in_order():
    rng = np.random.default_rng(2)
    lens = rng.integers(16000, 480001, size=500)
    waves = corpus.plan_waves(np.arange(500), lens, 4_000_000)
    assert waves[0][0] == 0 and sum(c for _, c in waves) == 500
    for (f, c), (f2, _) in zip(waves, waves[1:]):
        assert f + c == f2
    for f, c in waves:
        assert c >= 1 and (c == 1 or lens[f:f + c].sum() <= 4_000_000)
    assert corpus.plan_waves(np.arange(3), [10, 10, 10], 5) == [(0, 1), (1, 1), (2, 1)]      # an utterance above the budget is a wave of its own
    assert corpus.plan_waves([], [], 5) == []


def test_kill_and_resume_runs_each_wave_once(tmp_path):
    """a run killed after three waves leaves three manifest lines; the second run does the other waves only; a third run
    does nothing; a wave whose output has vanished is done again; a manifest of another corpus is refused"""
    waves = corpus.plan_waves(np.arange(10), [4] * 10, 8)
    assert len(waves) == 5
    sig = corpus._signature({"kind": "synthetic", "seed": 1}, "mfcc", dict(B=40), 0, 1, 8)
    path = str(tmp_path / "out" / "manifest.0.jsonl")
    calls = []

    class Killed(Exception):
        pass

    def do_wave(w, first, count, die_at=None):
        if die_at is not None and w == die_at:
            raise Killed()
        calls.append(w)
        (tmp_path / ("w%d" % w)).write_text("x")
        return {"ark": "w%d" % w, "frames": count}

    man = corpus.Manifest(path, sig, header={"rank": 0})
    with pytest.raises(Killed):
        corpus.walk(waves, man, lambda w, f, c: do_wave(w, f, c, die_at=3))
    assert calls == [0, 1, 2]
    lines = open(path).read().strip().split("\n")
    assert len(lines) == 4 and json.loads(lines[0])["signature"] == sig
    # a line cut short by the kill is ignored (that wave is redone)
    with open(path, "a") as f:
        f.write('{"wave": 3, "first": 6, "cou')
    exists = lambda rec: os.path.exists(str(tmp_path / rec["ark"]))        # noqa: E731
    man = corpus.Manifest(path, sig)
    assert sorted(man.done) == [0, 1, 2]
    assert corpus.walk(waves, man, do_wave, exists) == (2, 3)
    assert calls == [0, 1, 2, 3, 4]
    assert corpus.walk(waves, corpus.Manifest(path, sig), do_wave, exists) == (0, 5)
    os.remove(str(tmp_path / "w1"))
    assert corpus.walk(waves, corpus.Manifest(path, sig), do_wave, exists) == (1, 4)
    assert calls[-1] == 1
    with pytest.raises(ValueError):
        corpus.Manifest(path, corpus._signature({"kind": "synthetic", "seed": 2}, "mfcc", dict(B=40), 0, 1, 8))
    # other options, another rank or another world size are other signatures
    assert len({corpus._signature({"s": 1}, "mfcc", dict(B=40), r, w, 8) for r, w in ((0, 1), (0, 2), (1, 2))} |
               {corpus._signature({"s": 1}, "mfcc", dict(B=41), 0, 1, 8), corpus._signature({"s": 1}, "fbank", dict(B=40), 0, 1, 8)}) == 5


@pytest.mark.gpu
def test_device_generator_equals_the_host_generator():
    import torch
    from speech_amd import api
    lens = [16000, 1, 4097, 70000, 4096]
    ids = [3, 2 ** 33 + 1, 0, 99, 7]
    for dtype, tdt in (("f32", torch.float32), ("f64", torch.float64)):
        plan = api.Plan("mfcc", dtype=dtype, snip=0)
        b = api.Batch(plan, lens)
        for pcm in (False, True):
            X = torch.full((b.total_samples + 5,), 77, device="cuda", dtype=torch.int16 if pcm else tdt)
            b.synth_audio(X, seed=12345, ids=ids)
            torch.cuda.synchronize()
            x = X.cpu().numpy()
            assert (x[b.total_samples:] == 77).all()
            for u, (n, i) in enumerate(zip(lens, ids)):
                assert np.array_equal(x[int(b.x_off[u]):int(b.x_off[u]) + n].astype(np.int16), corpus.synth_utterance(12345, i, n)), (dtype, pcm, u)
        X = torch.zeros(b.total_samples, device="cuda", dtype=tdt)
        b.synth_audio(X)                                                   # default ids: 0, 1, 2, ...
        assert np.array_equal(X.cpu().numpy()[:16000].astype(np.int16), corpus.synth_utterance(0x5EED2026, 0, 16000))


@pytest.mark.gpu
def test_corpus_waves_resume_and_match_the_oracle(tmp_path):
    """0.2 h of hires MFCC in small waves, stopped after two waves and resumed: every utterance appears once in the
    archives, the resumed run skips the finished waves, and sampled utterances match the oracle on the host generator's audio"""
    import oracle_lib as O
    from compare import check_parity
    import np_truth as T
    from speech_amd import ark
    spec = corpus.SyntheticCorpus(hours=0.2)
    opts = dict(B=40, C=40, Q=22.0, hif=7600.0)
    out = str(tmp_path / "feats")
    os.makedirs(out)
    r1 = corpus.run_synthetic(spec, "mfcc", opts, wave_audio_s=100.0, out_dir=out, max_waves=2)
    assert r1["waves_run"] == 2 and r1["waves_skipped"] == 0
    r2 = corpus.run_synthetic(spec, "mfcc", opts, wave_audio_s=100.0, out_dir=out)
    assert r2["waves_skipped"] == 2 and r2["waves_run"] == r2["waves"] - 2 and r2["waves"] >= 5
    r3 = corpus.run_synthetic(spec, "mfcc", opts, wave_audio_s=100.0, out_dir=out)
    assert r3["waves_run"] == 0
    # totals are whole-corpus figures whatever was run now (from the manifest's records); *_run_now says what this call did
    total = int(spec.lens.sum())
    assert r1["samples"] == r1["samples_run_now"] < total
    assert r2["samples"] == r3["samples"] == total and r2["samples_run_now"] == total - r1["samples"] and r3["samples_run_now"] == 0
    assert r3["utterances"] == spec.lens.size and r3["kernel_ms"] > 0.0 and r3["kernel_ms_run_now"] == 0.0
    assert abs(r3["audio_s"] - total / 16000.0) < 1e-6 and r3["audio_s_run_now"] == 0.0
    with pytest.raises(ValueError):
        corpus.run_synthetic(spec, "mfcc", dict(opts, B=41), wave_audio_s=100.0, out_dir=out)
    feats = {}
    for rec in r3["records"]:
        for key, _, m in ark.read_ark(os.path.join(out, rec["ark"])):
            assert key not in feats
            feats[key] = m
    assert sorted(feats) == ["utt%09d" % u for u in range(spec.lens.size)]
    for u in (0, 1, spec.lens.size - 1):
        x = spec.audio(u, np.float32)
        got = feats["utt%09d" % u]
        assert got.shape == (1 + (x.size - 400) // 160, 40)
        _, ref = O.run("mfcc", x, **opts)
        _, ref_fast = O.run("mfcc", x, exact=False, **opts)
        res = check_parity("mfcc", opts, got, ref, T.mfcc(x.astype(np.float64), single=True, **opts), tol=1e-4, oracle_fast=ref_fast)
        assert res[0], res
    # two ranks: their waves together hold the same features (the audio of an utterance does not depend on the sharding)
    both = {}
    for rank in (0, 1):
        rr = corpus.run_synthetic(spec, "mfcc", opts, wave_audio_s=150.0, out_dir=str(tmp_path / "two"), rank=rank, world=2)
        for rec in rr["records"]:
            for key, _, m in ark.read_ark(os.path.join(str(tmp_path / "two"), rec["ark"])):
                assert key not in both
                both[key] = m
    assert sorted(both) == sorted(feats) and all(np.array_equal(both[k], feats[k]) for k in feats)


def test_file_corpus_streams_in_waves_and_resumes(tmp_path):
    """corpus.run_files with a stand-in extractor: waves bounded by file size, one archive + script file per wave, per-feature
    sums carried in the manifest, a killed run resumes after its last recorded wave and a changed option set is refused"""
    import wave as wavmod
    from speech_amd import ark, corpus
    rng = np.random.default_rng(5)
    keys, paths, lens = [], [], [3000, 500, 2500, 800, 4000, 100, 1500]
    for i, n in enumerate(lens):
        p = tmp_path / ("u%d.wav" % i)
        with wavmod.open(str(p), "wb") as w:
            w.setnchannels(1); w.setsampwidth(2); w.setframerate(8000)
            w.writeframes(rng.integers(-3000, 3000, n).astype(np.int16).tobytes())
        keys.append("utt%d" % i); paths.append(str(p))
    calls = []

    def extract(xs, ids, rate):
        calls.append([int(i) for i in ids])
        assert rate == 8000
        return [np.stack([x[:len(x) // 100 * 100].reshape(-1, 100).mean(1), np.full(len(x) // 100, float(i))], 1).astype(np.float32) for x, i in zip(xs, ids)]

    out = tmp_path / "out"
    res = corpus.run_files(keys, paths, "mfcc", {"B": 23}, str(out), wave_bytes=9000, extract=extract)
    assert res["waves"] == len(calls) >= 3 and res["waves_run"] == res["waves"] and res["utterances"] == 7
    assert all(sum(os.path.getsize(paths[i]) for i in c) <= 9000 or len(c) == 1 for c in calls)     # a wave holds at most wave_bytes of files
    got = {}
    for w in range(res["waves"]):
        got.update({k: m for k, m in ark.read_scp(str(out / ("feats.0.%d.scp" % w)))})                # script files point at the final archive names
    assert set(got) == set(keys) and res["frames"] == sum(m.shape[0] for m in got.values())
    allf = np.concatenate([got[k] for k in keys]).astype(np.float64)
    assert np.allclose(res["sum"], allf.sum(0)) and np.allclose(res["sumsq"], (allf * allf).sum(0))
    # resume: nothing to do; after losing the last wave's archive and a torn manifest line, only that wave is redone
    n0 = len(calls)
    res2 = corpus.run_files(keys, paths, "mfcc", {"B": 23}, str(out), wave_bytes=9000, extract=extract)
    assert len(calls) == n0 and res2["waves_skipped"] == res["waves"] and np.allclose(res2["sum"], res["sum"])
    os.remove(out / ("feats.0.%d.ark" % (res["waves"] - 1)))
    with open(out / "manifest.0.jsonl", "a") as f:
        f.write('{"wave": 99, "fir')
    res3 = corpus.run_files(keys, paths, "mfcc", {"B": 23}, str(out), wave_bytes=9000, extract=extract)
    assert len(calls) == n0 + 1 and res3["waves_run"] == 1 and np.allclose(res3["sum"], res["sum"])
    with pytest.raises(ValueError):
        corpus.run_files(keys, paths, "mfcc", {"B": 40}, str(out), wave_bytes=9000, extract=extract)
    # two ranks: disjoint shares, together the corpus
    r0 = corpus.run_files(keys, paths, "mfcc", {"B": 23}, str(tmp_path / "two"), rank=0, world=2, wave_bytes=9000, extract=extract)
    r1 = corpus.run_files(keys, paths, "mfcc", {"B": 23}, str(tmp_path / "two"), rank=1, world=2, wave_bytes=9000, extract=extract)
    assert r0["utterances"] + r1["utterances"] == 7 and np.allclose(r0["sum"] + r1["sum"], res["sum"])


def test_apply_cmvn_can_be_killed_and_repeated(tmp_path):
    """round-2 advisor finding: --apply-cmvn used to rewrite the extracted archives in place, so a run killed in that loop and
    started again normalised the already rewritten ones twice.  The normalised features now go to feats_cmvn.* beside the
    archives, always computed from the untouched extracted ones: kill after the first archive, run again, every normalised
    archive is x * scale + offset exactly once"""
    from speech_amd import ark
    rng = np.random.default_rng(4)
    D = 5
    raw, arks = {}, []
    for w in range(3):
        feats = [("utt%d_%d" % (w, i), rng.standard_normal((int(rng.integers(3, 9)), D)).astype(np.float32) * 10 + 3) for i in range(4)]
        path = str(tmp_path / ("feats.0.%d.ark" % w))
        ark.write_ark(path, feats)
        raw.update(dict(feats))
        arks.append(path)
    before = {p: open(p, "rb").read() for p in arks}
    scale, offset = rng.uniform(0.5, 2.0, D), rng.uniform(-1.0, 1.0, D)

    class Killed(Exception):
        pass

    def kill_after_first(i):
        if i == 0:
            raise Killed()

    with pytest.raises(Killed):
        corpus.apply_cmvn_to_archives(arks, scale, offset, _after_each=kill_after_first)
    assert os.path.exists(tmp_path / "feats_cmvn.0.0.ark") and not os.path.exists(tmp_path / "feats_cmvn.0.1.ark")
    out = corpus.apply_cmvn_to_archives(arks, scale, offset)            # the resumed run
    out2 = corpus.apply_cmvn_to_archives(arks, scale, offset)           # and once more for good measure
    assert out == out2 == [str(tmp_path / ("feats_cmvn.0.%d.ark" % w)) for w in range(3)]
    assert {p: open(p, "rb").read() for p in arks} == before            # the extracted archives were never touched
    seen = {}
    for p in out:
        for (k, _, m), (k2, m2) in zip(ark.read_ark(p), ark.read_scp(p[:-4] + ".scp")):
            assert k == k2 and np.array_equal(m, m2)
            seen[k] = m
    assert set(seen) == set(raw)
    for k, m in seen.items():
        want = (raw[k] * scale.astype(np.float32)).astype(np.float32) + offset.astype(np.float32)
        assert np.array_equal(m, want), k
    assert not [f for f in os.listdir(tmp_path) if f.endswith(".tmp")]
