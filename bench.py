#!/usr/bin/env python3
"""bench.py -- throughput of the feature-extraction hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One "step" is one pass of the per-frame pipeline (c/kaldi_fbank.c:208-351) over one resident batch of synthetic 16 kHz
audio.  The headline workload is the one BASELINE.json's metric is quoted on, configs[2]: kaldi_fbank, 80 log-mel bins,
over a ragged batch of 100,000 utterances of 1-30 s per GPU (99 GB of audio + 50 GB of features, resident in HBM).
The audio is the synthetic corpus of SURVEY.md 8(d) (integer counter-based generator, speech_batch_synth_audio: generated on
the device, bit-identical to the host statement, so the CPU legs regenerate the very samples instead of copying them back).
Utterances are independent, so ranks shard them with no collective ("weak" scaling: every rank owns 100,000 utterances,
its part of an LPT assignment of N x 100,000); the only cross-rank steps are the barrier, the max over ranks of the
elapsed time and the gather of the per-GPU figures -- bookkeeping on CPU tensors over gloo, in rehearsals and real runs alike
(RCCL is not linked: nothing on the data path has an exchange step).  After the timed region the output it left is CHECKED
(finite, last-bin quirk, three utterances against the CPU oracle) and the verdict is part of the line.  Started WITHOUT a launcher and with --gpus N > 1 it starts the N
ranks itself (fresh child processes, before the parent touches any GPU).  Rank 0 prints ONE JSON line; DESIGN.md
section "Measurement" explains every field.
"""
import argparse
import ctypes as C
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured float4 copy)
SR = 16000.0

FRONT = dict(snip=1, dc0=1, raw_energy=1, preemph=0.97, win="povey")
FBANK80 = dict(B=80, lg=1, lof=20.0, hif=8000.0, **FRONT)
MFCC13 = dict(B=23, C=13, Q=22.0, lof=20.0, hif=8000.0, **FRONT)
HIRES = dict(B=40, C=40, Q=22.0, lof=20.0, hif=7600.0, **FRONT)
PLP13 = dict(B=23, C=13, Q=22.0, compress=1.0 / 3.0, cep_scale=1.0, lof=20.0, hif=8000.0, **FRONT)
WORKLOADS = {
    # BASELINE.json configs; the headline is configs[2], the workload the metric names
    "fbank80_100k_ragged_1-30s": dict(kind="fbank", opts=FBANK80, lens="ragged", nutt=100000),
    "spectrogram_1000x10s": dict(kind="spectrogram", opts=dict(**FRONT), lens="fixed", nutt=1000, secs=10.0),
    "mfcc13_1000x10s": dict(kind="mfcc", opts=MFCC13, lens="fixed", nutt=1000, secs=10.0),
    "mfcc_hires40_1000x10s": dict(kind="mfcc", opts=HIRES, lens="fixed", nutt=1000, secs=10.0),
    "plp13_1000x10s": dict(kind="plp", opts=PLP13, lens="fixed", nutt=1000, secs=10.0),
    "fbank80_ragged_1-30s": dict(kind="fbank", opts=FBANK80, lens="ragged", nutt=2000),
    # the metric's other extractor (MFCC) and BASELINE.json configs[4] (PLP) at the headline's size: the same 100 000 ragged utterances
    "mfcc13_100k_ragged_1-30s": dict(kind="mfcc", opts=MFCC13, lens="ragged", nutt=100000),
    "plp13_100k_ragged_1-30s": dict(kind="plp", opts=PLP13, lens="ragged", nutt=100000),
    # widening: the same spectrogram batch fed as 16-bit PCM (speech_batch_run_pcm16), 2 N + 4 W D bytes
    "spectrogram_pcm16_1000x10s": dict(kind="spectrogram", opts=dict(**FRONT), lens="fixed", nutt=1000, secs=10.0, pcm16=True),
    "mfcc13_dither1_1000x10s": dict(kind="mfcc", opts=dict(MFCC13, dither=1.0), lens="fixed", nutt=1000, secs=10.0),
    "mfcc13_8kHz_1000x10s": dict(kind="mfcc", opts=dict(MFCC13, sr=8000.0, hif=4000.0), lens="fixed", nutt=1000, secs=10.0, sr=8000.0),
    "mfcc13_8kHz_dither1_1000x10s": dict(kind="mfcc", opts=dict(MFCC13, sr=8000.0, hif=4000.0, dither=1.0), lens="fixed", nutt=1000, secs=10.0, sr=8000.0),
    "spectrogram_8kHz_1000x10s": dict(kind="spectrogram", opts=dict(sr=8000.0, **FRONT), lens="fixed", nutt=1000, secs=10.0, sr=8000.0),
    # 1024-point frames (22.05 kHz audio, 551-sample frames): one frame per 32 lanes
    "fbank80_22kHz_1000x10s": dict(kind="fbank", opts=dict(FBANK80, sr=22050.0, hif=11025.0), lens="fixed", nutt=1000, secs=10.0, sr=22050.0),
    "spectrogram_22kHz_1000x10s": dict(kind="spectrogram", opts=dict(sr=22050.0, **FRONT), lens="fixed", nutt=1000, secs=10.0, sr=22050.0),
    "mfcc13_22kHz_dither1_1000x10s": dict(kind="mfcc", opts=dict(MFCC13, sr=22050.0, hif=11025.0, dither=1.0), lens="fixed", nutt=1000, secs=10.0, sr=22050.0),
    "fbank80_48kHz_1000x10s": dict(kind="fbank", opts=dict(FBANK80, sr=48000.0, hif=24000.0), lens="fixed", nutt=1000, secs=10.0, sr=48000.0),
    "mfcc13_48kHz_1000x10s": dict(kind="mfcc", opts=dict(MFCC13, sr=48000.0, hif=24000.0), lens="fixed", nutt=1000, secs=10.0, sr=48000.0),
    "fbank80_44kHz_1000x10s": dict(kind="fbank", opts=dict(FBANK80, sr=44100.0, hif=22050.0), lens="fixed", nutt=1000, secs=10.0, sr=44100.0),
    "mfcc13_48kHz_dither1_1000x10s": dict(kind="mfcc", opts=dict(MFCC13, sr=48000.0, hif=24000.0, dither=1.0), lens="fixed", nutt=1000, secs=10.0, sr=48000.0),
    "mfcc13_44kHz_1000x10s": dict(kind="mfcc", opts=dict(MFCC13, sr=44100.0, hif=22050.0), lens="fixed", nutt=1000, secs=10.0, sr=44100.0),
    "spectrogram_48kHz_1000x10s": dict(kind="spectrogram", opts=dict(sr=48000.0, **FRONT), lens="fixed", nutt=1000, secs=10.0, sr=48000.0),
    "mfcc13_22kHz_1000x10s": dict(kind="mfcc", opts=dict(MFCC13, sr=22050.0, hif=11025.0), lens="fixed", nutt=1000, secs=10.0, sr=22050.0),
}
HEADLINE = "fbank80_100k_ragged_1-30s"
HEADLINE_TEXT = ("kaldi_fbank, 80 log-mel bins, on a ragged batch of 100,000 16 kHz utterances of 1-30 s per GPU, 25/10 ms frames, "
                 "512-pt FFT (BASELINE.json configs[2])")
WAVE_AUDIO_S = 1.0e6           # corpus leg: audio-seconds per wave (64 GB of float32 audio + 16 GB of hires features in HBM)
COUNTERS_JSON = os.path.join(ROOT, "profiles", "r3_counters.json")       # written by tools/pmc_to_json.py from rocprofv3 --pmc passes
CORPUS_SEED = 0x5EED2026       # master seed of the synthetic corpus
FP32_VALU_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: dense f32 vector (and f32 matrix) peak
ALG_FLOP_PER_FRAME = {"spectrogram": 2000 + 11520 + 800, "fbank": 2000 + 11520 + 800 + 1000, "mfcc": 2000 + 11520 + 800 + 1000,
                      "plp": 2000 + 11520 + 800 + 1000}     # front end + real FFT-512 (2.5 n log2 n) + power + sparse mel; the DCT is added per plan (2 B C)


def lengths(w, rank, world, nutt):
    """(sample counts, corpus numbers) of this rank's utterances.  The corpus number keys the synthetic generator (and the
    dither), so an utterance's audio does not depend on which rank holds it."""
    if w["lens"] == "fixed":
        return np.full(nutt, int(w["secs"] * w.get("sr", SR)), dtype=np.uint64), np.arange(rank * nutt, (rank + 1) * nutt, dtype=np.uint64)
    # 1-30 s, SURVEY.md 8(d) config 3.  N ranks: one list of N x nutt lengths, dealt to the ranks by api.shard_utterances
    # (longest first to the least loaded rank; no utterance is split) -- every rank derives the same assignment by itself
    rng = np.random.default_rng(CORPUS_SEED)
    allens = rng.integers(16000, 480001, size=nutt * world).astype(np.int64)
    if world == 1:
        return allens.astype(np.uint64), np.arange(nutt, dtype=np.uint64)
    from speech_amd import api
    mine = api.shard_utterances(allens, world)[rank]
    return allens[mine].astype(np.uint64), mine.astype(np.uint64)


def host_audio(ids, lens):
    """the same samples on the host: speech_synth_audio_host, the C statement of the generator (bit-identical to the device's,
    tests/test_corpus.py) -- nothing is copied back from the GPU for the CPU legs"""
    from speech_amd import _lib
    lib = _lib.load()
    out = []
    for u, n in zip(ids, lens):
        a = np.empty(int(n), dtype=np.int16)
        assert lib.speech_synth_audio_host(CORPUS_SEED, int(u), int(n), a.ctypes.data_as(C.POINTER(C.c_short))) == 0
        out.append(a)
    return out


class Work:
    def __init__(self, name, device, rank=0, world=1, nutt=None):
        import torch
        from speech_amd import api
        self.name = name
        w = WORKLOADS[name]
        self.kind = w["kind"]
        self.opts = w["opts"]
        self.sr = w.get("sr", SR)
        self.plan = api.Plan(w["kind"], device=device.index, **w["opts"])
        self.lens, self.ids = lengths(w, rank, world, nutt or w["nutt"])
        self.batch = api.Batch(self.plan, self.lens)
        self.pcm16 = bool(w.get("pcm16"))
        self.X = torch.empty(self.batch.total_samples, device=device, dtype=torch.int16 if self.pcm16 else torch.float32)
        self.batch.synth_audio(self.X, seed=CORPUS_SEED, ids=self.ids, stream=torch.cuda.current_stream().cuda_stream)
        self.Y = torch.zeros(self.batch.total_out, device=device, dtype=torch.float32)
        self.audio_s = float(self.batch.total_samples) / self.sr
        # algorithmic bytes per launch: read every sample once, write every feature once
        self.bytes = (2.0 if self.pcm16 else 4.0) * self.batch.total_samples + 4.0 * self.batch.total_out

    def step(self, stream):
        self.batch.run(self.X, self.Y, stream=stream)

    def kernel_ms(self, stream, reps=5):
        """isolated launches, each bracketed by HIP events inside the library (speech_batch_run_timed) and waited for"""
        ms = [self.batch.run(self.X, self.Y, stream=stream, timed=True) for _ in range(reps)]
        return float(np.mean(ms)), float(np.min(ms))

    def sustained_ms(self, steps):
        """average launch duration of `steps` back-to-back launches, HIP events on the launch stream (torch's current one)"""
        import torch
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(steps):
            self.batch.run(self.X, self.Y, stream=torch.cuda.current_stream().cuda_stream)
        e1.record()
        e1.synchronize()
        return e0.elapsed_time(e1) / steps

    def free(self):
        self.batch.close()
        self.plan.close()
        self.X = self.Y = None


def cpu_baseline(work, budget_s=10.0):
    """The CPU oracle (oracle/, fast-FFT mode, -O3) timed on this host over a bounded sample of the same workload."""
    so = os.path.join(ROOT, "oracle", "liboracle_native.so")
    if not os.path.exists(so):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    lib = O.lib("liboracle_native.so")
    lib.oracle_set_fft_exact(0)
    threads = int(os.environ.get("OMP_NUM_THREADS", "0")) or min(16, os.cpu_count() or 1)
    o = work.opts
    kind_id = {"spectrogram": 0, "fbank": 1, "mfcc": 2, "plp": 3}[work.kind]
    pz, pf = C.POINTER(C.c_size_t), C.POINTER(C.c_float)

    cache = {}

    def run(nutt):
        lens = work.lens[:nutt].astype(np.uint64)
        xo = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.uint64)
        nf = work.batch.n_frames[:nutt].astype(np.uint64)
        yo = np.concatenate([[0], np.cumsum(nf * work.plan.D)[:-1]]).astype(np.uint64)
        if nutt not in cache:                                            # regenerated on the host, once, outside the timed call
            cache[nutt] = np.concatenate(host_audio(work.ids[:nutt], lens)).astype(np.float32)
        X = cache[nutt]
        Y = np.zeros(int((nf * work.plan.D).sum()), dtype=np.float32)
        t0 = time.perf_counter()
        used = lib.oracle_batch_s(kind_id, nutt, X.ctypes.data_as(pf), xo.ctypes.data_as(pz), lens.ctypes.data_as(pz), Y.ctypes.data_as(pf),
                                  yo.ctypes.data_as(pz), SR, 25.0, 10.0, o["snip"], o["dc0"], o["raw_energy"], o["preemph"], o["win"].encode(),
                                  0, o.get("lof", 20.0), o.get("hif", 8000.0), o.get("B", 23), o.get("lg", 1), o.get("C", 13), o.get("Q", 22.0),
                                  o.get("compress", 1.0 / 3.0), o.get("cep_scale", 1.0), 0, threads)
        dt = time.perf_counter() - t0
        assert used >= 1, "oracle batch failed"
        return dt, float(lens.sum()) / SR, used

    n = int(min(len(work.lens), 16 * threads))
    run(min(n, 2 * threads))                                            # warm-up: page in the library, spin up the OpenMP team
    tot_dt = tot_secs = 0.0
    passes = 0
    while tot_dt < budget_s and passes < 1000:                          # repeat the bounded sample until ~budget_s of CPU wall time
        dt, secs, used = run(n)
        tot_dt += dt; tot_secs += secs; passes += 1
    return {"value": tot_secs / tot_dt, "unit": "audio-s/s", "cores": used, "kind": "port",
            "sample": "%d passes over the first %d utterances (%.0f audio-s in all) of %s, oracle/liboracle_native.so (C restatement of the "
                      "reference's per-utterance call, float FFT, OpenMP over utterances) on the same samples regenerated on the host, %.1f s wall"
                      % (passes, n, tot_secs, work.name, tot_dt)}


def config0_single_utterance():
    """BASELINE.json configs[0]: kaldi_mfcc on ONE 10 s utterance through the drop-in symbol (host pointers, synchronous:
    H2D + kernel + D2H per call), beside the same call of the CPU oracle on one host thread."""
    from speech_amd import api
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    rng = np.random.default_rng(7)
    x = (1000.0 * rng.standard_normal(160000)).astype(np.float32)
    api.kaldi_mfcc(x)
    t0 = time.perf_counter()
    for _ in range(50):
        api.kaldi_mfcc(x)
    gpu_ms = (time.perf_counter() - t0) / 50 * 1e3
    lib = O.lib("liboracle_native.so")
    lib.oracle_set_fft_exact(0)
    pz, pf = C.POINTER(C.c_size_t), C.POINTER(C.c_float)
    y = np.zeros(998 * 13, dtype=np.float32)
    zero, n = np.zeros(1, dtype=np.uint64), np.array([x.size], dtype=np.uint64)

    def cpu_call():
        assert lib.oracle_batch_s(2, 1, x.ctypes.data_as(pf), zero.ctypes.data_as(pz), n.ctypes.data_as(pz), y.ctypes.data_as(pf), zero.ctypes.data_as(pz),
                                  SR, 25.0, 10.0, 1, 1, 1, 0.97, b"povey", 0, 20.0, 8000.0, 23, 1, 13, 22.0, 1.0 / 3.0, 1.0, 0, 1) >= 1
    cpu_call()
    t0 = time.perf_counter()
    for _ in range(20):
        cpu_call()
    cpu_ms = (time.perf_counter() - t0) / 20 * 1e3
    return {"workload": "kaldi_mfcc_s, one 10 s utterance, 23 mel / 13 ceps, host pointers in and out",
            "hip_ms_per_call": gpu_ms, "hip_audio_s_per_s": 10.0 / (gpu_ms * 1e-3),
            "cpu_oracle_ms_per_call_1_thread": cpu_ms, "cpu_audio_s_per_s": 10.0 / (cpu_ms * 1e-3)}


def pcie_inclusive(work, device, nutt=64):
    """Host buffers in, host buffers out (what the drop-in kaldi_*_s symbols do per call), batched: H2D + kernels + D2H."""
    import torch
    from speech_amd import api
    lens = work.lens[:nutt]
    b = api.Batch(work.plan, lens)
    Xh = torch.empty(b.total_samples, dtype=torch.float32).pin_memory()
    Xh.copy_(torch.from_numpy(np.concatenate(host_audio(work.ids[:nutt], lens)).astype(np.float32)))
    Yh = torch.empty(b.total_out, dtype=torch.float32).pin_memory()
    Xd = torch.empty(b.total_samples, device=device)
    Yd = torch.empty(b.total_out, device=device)
    s = torch.cuda.current_stream().cuda_stream
    for it in range(3):
        if it == 1:
            torch.cuda.synchronize(); t0 = time.perf_counter()
        Xd.copy_(Xh, non_blocking=True)
        b.run(Xd, Yd, stream=s)
        Yh.copy_(Yd, non_blocking=True)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 2
    b.close()
    return float(b.total_samples) / SR / dt


def corpus_waves(device, hours, steps_note=None):
    """BASELINE.json configs[3] on one GPU: hires MFCC (40 mel / 40 ceps) over `hours` of synthetic audio walked in HBM-sized
    waves (speech_amd.corpus), every wave generated on the device by the integer generator and never materialised on the
    host; only the extraction launches are timed (HIP events), the generator beside them."""
    from speech_amd import corpus
    spec = corpus.SyntheticCorpus(hours=hours, seed=0x5EED2026)
    r = corpus.run_synthetic(spec, "mfcc", HIRES, device=device.index, wave_audio_s=WAVE_AUDIO_S, keep_output=False, pcm16=False)
    by = 4.0 * r["samples"] + 4.0 * r["features"]
    return {"workload": "kaldi_mfcc hires (40 mel / 40 ceps) on %.0f h of synthetic 16 kHz audio = 1/%d of BASELINE.json configs[3]'s 10k-hour corpus, "
                        "%d waves of <= %.0f audio-s (float32, resident) generated on the device" % (hours, round(10000.0 / hours), r["waves"], WAVE_AUDIO_S),
            "audio_s": r["audio_s"], "utterances": r["utterances"], "extract_ms_total": r["kernel_ms"], "generate_ms_total": r["generate_ms"],
            "audio_s_per_s_per_gpu": r["audio_s"] / (r["kernel_ms"] * 1e-3), "roofline_frac_hbm": by / (r["kernel_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "wall_s_incl_generation": r["wall_s"], "ten_k_hours_on_8_gpus_s_extrapolated": r["kernel_ms"] * 1e-3 * (10000.0 / hours) / 8.0}


def utilisation(workload):
    """VALU / MFMA utilisation and HBM traffic of the workload's dominant kernel from the committed rocprofv3 --pmc passes
    (profiles/r3_counters.json, produced by tools/gpu_prof.sh + tools/pmc_to_json.py on the same bench command).  The record
    names the sources it was measured on (kernel_sources_sha, commit); when the library this run loads was built from other
    kernel sources the counters are NOT reported (the caller nulls them and sets counters_stale)."""
    try:
        u = json.load(open(COUNTERS_JSON)).get(workload, {})
    except Exception:
        return {}
    from speech_amd.srchash import kernel_sources_sha
    u = dict(u)
    u["stale"] = bool(u) and u.get("kernel_sources_sha") != kernel_sources_sha()
    return u


def counter_fields(u):
    live = bool(u) and not u.get("stale")
    pick = lambda k: (u.get(k) if live else None)                       # noqa: E731
    return {"traffic": pick("hbm_bytes_per_launch"), "valu_util": pick("valu_util"), "mfma_util": pick("mfma_util"),
            "lds_bank_conflict_frac": pick("lds_bank_conflict_frac"), "counters_from": u.get("source"), "counters_commit": u.get("commit"),
            "counters_kernel_sources_sha": u.get("kernel_sources_sha"), "counters_stale": bool(u.get("stale")) if u else None}


def output_check(work, picks=3):
    """What the timed region left in work.Y, checked: every feature finite, the reference's last-bin / last-cepstrum / Nyquist
    quirk on every frame (c/kaldi_fbank.c:350, c/kaldi_mfcc.c:333, c/kaldi_spectrogram.c:244-248), and `picks` utterances (the
    first, the shortest and the longest of the batch) against the CPU oracle on the same samples regenerated on the host, under
    the parity rule of tests/compare.py.  The oracle is the checker here, never the thing timed."""
    import torch
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import np_truth as T
    import oracle_lib as O
    from compare import check_parity, PARITY_LOG
    D = int(work.plan.D)
    Yv = work.Y.view(-1, D)
    finite, quirk = True, True
    for i in range(0, Yv.shape[0], 1 << 24):
        blk = Yv[i:i + (1 << 24)]
        finite = finite and bool(torch.isfinite(blk).all())
        if D >= 2:
            quirk = quirk and bool((blk[:, D - 1] == blk[:, D - 2]).all())
    alias = {"snip": "snip", "win": "win"}
    o = {alias.get(k, k): v for k, v in work.opts.items()}
    truth = {"spectrogram": T.spectrogram, "fbank": T.fbank, "mfcc": T.mfcc, "plp": T.plp}[work.kind]
    tol = 1e-5 if work.kind == "spectrogram" else 1e-4
    us = sorted({0, int(np.argmin(work.lens)), int(np.argmax(work.lens))})[:picks]
    if float(o.get("dither", 0.0)) > 0.0:
        us = []                                                        # dithered plans draw their own noise: no oracle to compare with
    rows, ok = [], True
    for u in us:
        x = host_audio(work.ids[u:u + 1], work.lens[u:u + 1])[0].astype(np.float32)
        W = int(work.batch.n_frames[u])
        got = work.Y[int(work.batch.y_off[u]):int(work.batch.y_off[u]) + W * D].cpu().numpy().reshape(W, D)
        _, ref = O.run(work.kind, x, single=True, **o)
        _, ref_fast = O.run(work.kind, x, single=True, exact=False, **o)
        good, e_go, e_gt, e_ot = check_parity(work.kind, o, got, ref, truth(x.astype(np.float64), single=True, **o), tol=tol, oracle_fast=ref_fast,
                                              label="bench")
        rows.append({"utt": int(work.ids[u]), "samples": int(work.lens[u]), "e_go": e_go, "hip_vs_f64": e_gt, "f32_cpu_vs_f64": e_ot,
                     "rms_noise_ratio": PARITY_LOG[-1][6], "clause": PARITY_LOG[-1][2]})
        ok = ok and good
    return {"finite": finite, "last_bin_quirk": quirk, "tol": tol, "vs_oracle": rows, "ok": bool(finite and quirk and ok)}


def visible_gpu_count():
    """GPUs this process would see, WITHOUT a HIP call (the launcher parent must not initialise the GPU: it forks the ranks):
    the *_VISIBLE_DEVICES lists if set, else the KFD topology nodes that have SIMDs.  None when neither can be read."""
    counts = []
    for var in ("HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            counts.append(len([t for t in v.split(",") if t.strip()]))
    if counts:
        return min(counts)
    n, seen = 0, False
    import glob
    for path in glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties"):
        try:
            for line in open(path):
                if line.startswith("simd_count"):
                    seen = True
                    n += 1 if int(line.split()[1]) > 0 else 0
        except OSError:
            pass
    return n if seen else None


def spawn_ranks(n, argv):
    """--gpus N without a launcher: start N fresh ranks of this script (the parent makes no GPU call), wait for them."""
    rehearsal = os.environ.get("BENCH_REHEARSAL") == "1" or os.environ.get("BENCH_LAUNCH_TEST") == "1"
    if not rehearsal:
        ndev = visible_gpu_count()                                     # sysfs / environment only: the parent makes no HIP call
        if ndev is not None and ndev < n:
            sys.stderr.write("bench.py: --gpus %d but only %d GPU(s) visible (BENCH_REHEARSAL=1 runs every rank on GPU 0)\n" % (n, ndev))
            return 2                                                   # (ndev unknown: every rank checks its own device below)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env))
    rc = 0
    deadline = time.time() + float(os.environ.get("BENCH_SPAWN_TIMEOUT", "1500"))
    pending = list(procs)
    while pending:
        for p in list(pending):
            code = p.poll()
            if code is not None:
                pending.remove(p)
                if code != 0:
                    rc = rc or code
        if (rc or time.time() > deadline) and pending:                  # a rank failed (or hung): stop the others, by PID
            for p in pending:
                p.kill()
            rc = rc or 124
            break
        time.sleep(0.05)
    for p in procs:
        p.wait()
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default=HEADLINE, choices=sorted(WORKLOADS))
    ap.add_argument("--nutt", type=int, default=0, help="utterances per GPU (default: the workload's own; rehearsals on one GPU need fewer)")
    ap.add_argument("--corpus-hours", type=float, default=1250.0, help="size of the wave-streamed hires-MFCC corpus leg (0 = skip)")
    ap.add_argument("--no-extras", action="store_true", help="skip cpu_baseline / other workloads (profiling runs)")
    ap.add_argument("--no-check", action="store_true", help="skip the check of the timed output (only for libspeech_diag.so runs whose kernels skip phases on purpose)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:]))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("WORLD_SIZE %d != --gpus %d" % (world, args.gpus))
    import torch
    import torch.distributed as dist
    # BENCH_REHEARSAL=1: every rank on GPU 0 with the gloo backend -- lets the N>1 code path be exercised on a one-GPU box.
    # BENCH_LAUNCH_TEST=1 (tests/test_bench_launcher.py, CPU): the ranks form the gloo group and report, without any GPU work.
    launch_test = os.environ.get("BENCH_LAUNCH_TEST") == "1"
    rehearsal = os.environ.get("BENCH_REHEARSAL") == "1" or launch_test
    if not rehearsal and local_rank >= torch.cuda.device_count():       # before the group forms: every rank of a node sees the same count
        sys.stderr.write("bench.py: --gpus %d but rank %d has no GPU %d (%d visible)\n" % (args.gpus, rank, local_rank, torch.cuda.device_count()))
        sys.exit(2)
    if world > 1:
        # ONE code path for rehearsals and real runs: the process group is gloo on CPU tensors.  It carries the barrier and the
        # gather of a few numbers per rank -- the data path has no exchange step and RCCL is not linked.
        # gloo's C++ side reports its connections on stdout ("[Gloo] Rank 0 is connected to ..."); stdout carries the ONE
        # JSON line and nothing else, so file descriptor 1 points at stderr while the group forms
        sys.stdout.flush()
        saved = os.dup(1)
        os.dup2(2, 1)
        try:
            dist.init_process_group("gloo")
            dist.barrier()
        finally:
            sys.stdout.flush()
            os.dup2(saved, 1)
            os.close(saved)
    if launch_test:
        t = torch.tensor([float(rank + 1)], dtype=torch.float64)
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dist.barrier()
            dist.destroy_process_group()
        if rank == 0:
            print(json.dumps({"launch_test": True, "n_gpus": world, "max_rank_plus_1": float(t.item())}))
        return
    # the only difference between a rehearsal and a real run: which device the rank selects
    device = torch.device("cuda", 0 if rehearsal else local_rank)
    torch.cuda.set_device(device)
    props = torch.cuda.get_device_properties(device)
    dev_info = {"name": props.name, "pci_bus_id": "%04x:%02x:%02x.0" % (getattr(props, "pci_domain_id", 0), getattr(props, "pci_bus_id", 0), getattr(props, "pci_device_id", 0)),
                "uuid": str(getattr(props, "uuid", "")), "hip_device": device.index, "pid": os.getpid()}

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    nutt = args.nutt or WORKLOADS[args.workload]["nutt"]
    if rehearsal and world > 1 and not args.nutt:
        nutt = max(1, nutt // (2 * world))                            # the ranks share ONE GPU's HBM
    work = Work(args.workload, device, rank, world, nutt)
    stream = torch.cuda.current_stream().cuda_stream
    for _ in range(args.warmup):
        work.step(stream)
    barrier()
    # HIP events on the launch stream bracket the same K launches the wall clock times: one launch per step, back to back,
    # so (event time) / K is the kernel's average launch duration over the timed region
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(args.steps):
        work.step(stream)
    ev1.record()
    barrier()
    elapsed = time.perf_counter() - t0
    k_avg = ev0.elapsed_time(ev1) / args.steps
    mine = [elapsed, work.audio_s, k_avg, float(work.bytes)]
    per, infos = [mine], [dev_info]
    if world > 1:
        t = torch.tensor(mine, dtype=torch.float64)
        allt = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(allt, t)                                       # bookkeeping only: 4 doubles per rank, CPU tensors
        per = [[float(v) for v in a] for a in allt]
        infos = [None] * world
        dist.all_gather_object(infos, dev_info)                        # which card each rank ran on
    elapsed = max(p[0] for p in per)                                   # MAX over ranks
    total_audio_s = sum(p[1] for p in per)

    out = None
    if rank == 0:
        k_iso, k_min = work.kernel_ms(stream)
        achieved = work.bytes / (k_avg * 1e-3) / 1e9
        u = utilisation(args.workload)
        # what the timed launches wrote, before anything overwrites it (skipped only on request, and the line says so)
        check = {"skipped": True, "ok": None, "vs_oracle": []} if args.no_check else output_check(work)
        # compute-side figure: SURVEY.md 8(d)'s algorithmic flop (sparse mel, real FFT as 2.5 n log2 n) against the f32 vector peak
        o_ = work.opts
        flop_frame = ALG_FLOP_PER_FRAME[work.kind] + (2.0 * o_.get("B", 23) * o_.get("C", 13) if work.kind == "mfcc" else 0.0)
        tflops = flop_frame * float(work.batch.total_frames) / (k_avg * 1e-3) / 1e12
        out = {
            "metric": "audio-sec processed/sec (RTF^-1), 16 kHz", "value": total_audio_s * args.steps / elapsed, "unit": "audio-s/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": HEADLINE_TEXT if args.workload == HEADLINE and nutt == WORKLOADS[HEADLINE]["nutt"] else "%s (%d utterances per GPU)" % (args.workload, nutt),
                       "utterances_per_gpu": int(len(work.lens)), "audio_s_per_gpu": work.audio_s, "frames_per_gpu": int(work.batch.total_frames),
                       "feature_dim": int(work.plan.D), "tuned_kernel": bool(work.plan.tuned), "rehearsal_on_one_gpu": bool(rehearsal and world > 1),
                       "sharding": "utterances over ranks (LPT on sample counts), no collective on the data path"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         **counter_fields(u),
                         "compute_side": {"algorithmic_flop_per_frame": flop_frame, "achieved_tflops": tflops, "peak_tflops_f32_valu": FP32_VALU_PEAK_TFLOPS,
                                          "frac": tflops / FP32_VALU_PEAK_TFLOPS,
                                          "nearer_bound": "valu" if tflops / FP32_VALU_PEAK_TFLOPS > achieved / HBM_PEAK_GBS else "hbm"},
                         "kernel_ms_avg": k_avg, "kernel_ms_isolated_avg": k_iso, "kernel_ms_isolated_min": k_min,
                         "algorithmic_bytes_per_launch": work.bytes},
            "output_check": check,
            "per_gpu": [{"rank": r, "audio_s_per_s": p[1] * args.steps / p[0], "kernel_ms_avg": p[2], "audio_s": p[1],
                         "roofline_frac_hbm": p[3] / (p[2] * 1e-3) / 1e9 / HBM_PEAK_GBS, "device": infos[r]} for r, p in enumerate(per)],
        }
        if check["ok"] is False:
            sys.stderr.write("bench.py: the timed output FAILED its check: %s\n" % json.dumps(check))
        if not args.no_extras and world == 1:       # the CPU leg runs on rank 0 at N=1 only
            out["cpu_baseline"] = cpu_baseline(work)
            out["pcie_inclusive_audio_s_per_s"] = pcie_inclusive(work, device)
            out["config0_single_utterance"] = config0_single_utterance()
    if not args.no_extras and world == 1 and args.workload == HEADLINE:
        others = {}
        work.free()
        torch.cuda.empty_cache()
        for name in WORKLOADS:
            if name == HEADLINE:
                continue
            w = Work(name, device)
            for _ in range(2):
                w.step(stream)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            kw = w.sustained_ms(args.steps)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / args.steps
            cf = counter_fields(utilisation(name))
            oc = output_check(w, picks=1)
            others[name] = {"audio_s_per_s_per_gpu": w.audio_s / dt, "kernel_ms_avg": kw, "roofline_frac_hbm": w.bytes / (kw * 1e-3) / 1e9 / HBM_PEAK_GBS,
                            "valu_util": cf["valu_util"], "mfma_util": cf["mfma_util"], "counters_stale": cf["counters_stale"], "tuned_kernel": bool(w.plan.tuned),
                            "output_ok": oc["ok"], "e_go": oc["vs_oracle"][0]["e_go"] if oc["vs_oracle"] else None}
            w.free()
            torch.cuda.empty_cache()
        out["other_workloads_1gpu"] = others
        if args.corpus_hours > 0:
            out["corpus_hires_mfcc_1gpu"] = corpus_waves(device, args.corpus_hours)
    if world > 1:
        barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))
        if out["output_check"]["ok"] is False:
            sys.exit(3)                                                # a bench line whose output failed its check is not a result


if __name__ == "__main__":
    main()
