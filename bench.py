#!/usr/bin/env python3
"""bench.py -- throughput of the feature-extraction hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One "step" is one pass of the per-frame pipeline over one resident batch of synthetic 16 kHz audio.  The headline
workload is BASELINE.json configs[1]: kaldi_spectrogram over 1,000 x 10 s utterances per GPU (inputs already in HBM,
outputs left in HBM).  Utterances are independent, so ranks shard them with no collective ("weak" scaling: every rank
owns 1,000 utterances); the only cross-rank step is the barrier + max over ranks of the elapsed time.
Rank 0 prints ONE JSON line; see DESIGN.md section "Measurement" for every field.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured float4 copy)
SR = 16000.0

FRONT = dict(snip=1, dc0=1, raw_energy=1, preemph=0.97, win="povey")
WORKLOADS = {
    # BASELINE.json configs; the headline is configs[1]
    "spectrogram_1000x10s": dict(kind="spectrogram", opts=dict(**FRONT), lens="fixed", nutt=1000, secs=10.0),
    "fbank80_ragged_1-30s": dict(kind="fbank", opts=dict(B=80, lg=1, lof=20.0, hif=8000.0, **FRONT), lens="ragged", nutt=2000),
    "mfcc13_1000x10s": dict(kind="mfcc", opts=dict(B=23, C=13, Q=22.0, lof=20.0, hif=8000.0, **FRONT), lens="fixed", nutt=1000, secs=10.0),
    "mfcc_hires40_1000x10s": dict(kind="mfcc", opts=dict(B=40, C=40, Q=22.0, lof=20.0, hif=7600.0, **FRONT), lens="fixed", nutt=1000, secs=10.0),
    # widening: the same spectrogram batch fed as 16-bit PCM (speech_batch_run_pcm16), 2 N + 4 W D bytes
    "spectrogram_pcm16_1000x10s": dict(kind="spectrogram", opts=dict(**FRONT), lens="fixed", nutt=1000, secs=10.0, pcm16=True),
    "mfcc13_dither1_1000x10s": dict(kind="mfcc", opts=dict(B=23, C=13, Q=22.0, lof=20.0, hif=8000.0, **{**FRONT, "dither": 1.0}), lens="fixed", nutt=1000, secs=10.0),
    "mfcc13_8kHz_1000x10s": dict(kind="mfcc", opts=dict(sr=8000.0, B=23, C=13, Q=22.0, lof=20.0, hif=4000.0, **FRONT), lens="fixed", nutt=1000, secs=10.0, sr=8000.0),
    "mfcc13_8kHz_dither1_1000x10s": dict(kind="mfcc", opts=dict(sr=8000.0, B=23, C=13, Q=22.0, lof=20.0, hif=4000.0, **{**FRONT, "dither": 1.0}), lens="fixed", nutt=1000, secs=10.0, sr=8000.0),
    "spectrogram_8kHz_1000x10s": dict(kind="spectrogram", opts=dict(sr=8000.0, **FRONT), lens="fixed", nutt=1000, secs=10.0, sr=8000.0),
    # 1024-point frames (22.05 kHz audio, 551-sample frames): one frame per 32 lanes
    "fbank80_22kHz_1000x10s": dict(kind="fbank", opts=dict(sr=22050.0, B=80, lg=1, lof=20.0, hif=11025.0, **FRONT), lens="fixed", nutt=1000, secs=10.0, sr=22050.0),
    "spectrogram_22kHz_1000x10s": dict(kind="spectrogram", opts=dict(sr=22050.0, **FRONT), lens="fixed", nutt=1000, secs=10.0, sr=22050.0),
    "mfcc13_22kHz_dither1_1000x10s": dict(kind="mfcc", opts=dict(sr=22050.0, B=23, C=13, Q=22.0, lof=20.0, hif=11025.0, **{**FRONT, "dither": 1.0}), lens="fixed", nutt=1000, secs=10.0, sr=22050.0),
    "mfcc13_22kHz_1000x10s": dict(kind="mfcc", opts=dict(sr=22050.0, B=23, C=13, Q=22.0, lof=20.0, hif=11025.0, **FRONT), lens="fixed", nutt=1000, secs=10.0, sr=22050.0),
    "plp13_1000x10s": dict(kind="plp", opts=dict(B=23, C=13, Q=22.0, compress=1.0 / 3.0, cep_scale=1.0, lof=20.0, hif=8000.0, **FRONT), lens="fixed", nutt=1000, secs=10.0),
}
HEADLINE = "spectrogram_1000x10s"


def lengths(w, rank):
    if w["lens"] == "fixed":
        return np.full(w["nutt"], int(w["secs"] * w.get("sr", SR)), dtype=np.uint64)
    rng = np.random.default_rng(0x5EED2026 + rank)
    return rng.integers(16000, 480001, size=w["nutt"]).astype(np.uint64)     # 1-30 s, SURVEY.md 8(d) config 3


def synth_audio(total, device, seed):
    """int16-scale noise plus slow tones, generated on the device (nothing crosses PCIe)."""
    import torch
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    x = torch.empty(total, device=device, dtype=torch.float32)
    chunk = 1 << 26
    for i in range(0, total, chunk):
        n = min(chunk, total - i)
        t = torch.arange(i, i + n, device=device, dtype=torch.float32)
        x[i:i + n] = (1000.0 * torch.randn(n, device=device, generator=g)
                      + 3000.0 * torch.sin(t * (2 * np.pi * 440.0 / SR)) + 1500.0 * torch.sin(t * (2 * np.pi * 2750.0 / SR)))
    return x


class Work:
    def __init__(self, name, device, rank):
        import torch
        from speech_amd import api
        self.name = name
        w = WORKLOADS[name]
        self.kind = w["kind"]
        self.opts = w["opts"]
        self.plan = api.Plan(w["kind"], device=device.index, **w["opts"])
        self.lens = lengths(w, rank)
        self.batch = api.Batch(self.plan, self.lens)
        self.X = synth_audio(self.batch.total_samples, device, 1234 + rank)
        self.pcm16 = bool(w.get("pcm16"))
        if self.pcm16:
            self.X = self.X.round().clamp(-32767, 32767).to(torch.int16)
        self.Y = torch.zeros(self.batch.total_out, device=device, dtype=torch.float32)
        self.audio_s = float(self.batch.total_samples) / w.get("sr", SR)
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
    import subprocess
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

    def run(nutt):
        lens = work.lens[:nutt].astype(np.uint64)
        xo = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.uint64)
        nf = work.batch.n_frames[:nutt].astype(np.uint64)
        yo = np.concatenate([[0], np.cumsum(nf * work.plan.D)[:-1]]).astype(np.uint64)
        X = work.X[:int(lens.sum())].cpu().numpy()
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
                      "reference's per-utterance call, float FFT, OpenMP over utterances), %.1f s wall"
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
    Xh.copy_(work.X[:b.total_samples].cpu())
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default=HEADLINE, choices=sorted(WORKLOADS))
    ap.add_argument("--no-extras", action="store_true", help="skip cpu_baseline / other workloads (profiling runs)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit("WORLD_SIZE %d != --gpus %d" % (world, args.gpus))
    # BENCH_REHEARSAL=1: every rank on GPU 0 with the gloo backend -- lets the N>1 code path be exercised on a one-GPU box
    rehearsal = os.environ.get("BENCH_REHEARSAL") == "1"
    device = torch.device("cuda", 0 if rehearsal else local_rank)
    torch.cuda.set_device(device)
    if world > 1:
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=device)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            if rehearsal:
                dist.barrier()
            else:
                dist.barrier(device_ids=[local_rank])
        torch.cuda.synchronize()

    work = Work(args.workload, device, rank)
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
    if world > 1:
        t = torch.tensor([elapsed], device="cpu" if rehearsal else device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    out = None
    if rank == 0:
        k_iso, k_min = work.kernel_ms(stream)
        achieved = work.bytes / (k_avg * 1e-3) / 1e9
        traffic = None
        tf = os.path.join(ROOT, "profiles", "hbm_traffic.json")      # written from rocprofv3 --pmc passes, see profiles/README.md
        if os.path.exists(tf):
            try:
                traffic = json.load(open(tf)).get(args.workload, {}).get("bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "audio-sec processed/sec (RTF^-1), 16 kHz", "value": world * work.audio_s * args.steps / elapsed, "unit": "audio-s/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "kaldi_spectrogram on 1,000 x 10 s 16 kHz utterances per GPU, 25/10 ms frames, 512-pt FFT power spectrum (BASELINE.json configs[1])"
                       if args.workload == HEADLINE else args.workload,
                       "utterances_per_gpu": int(len(work.lens)), "audio_s_per_gpu": work.audio_s, "frames_per_gpu": int(work.batch.total_frames),
                       "feature_dim": int(work.plan.D), "tuned_kernel": bool(work.plan.tuned), "sharding": "utterances over ranks, no collective"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "kernel_ms_avg": k_avg, "kernel_ms_isolated_avg": k_iso, "kernel_ms_isolated_min": k_min,
                         "algorithmic_bytes_per_launch": work.bytes},
        }
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
            w = Work(name, device, rank)
            for _ in range(2):
                w.step(stream)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            kw = w.sustained_ms(args.steps)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / args.steps
            others[name] = {"audio_s_per_s_per_gpu": w.audio_s / dt, "kernel_ms_avg": kw, "roofline_frac_hbm": w.bytes / (kw * 1e-3) / 1e9 / HBM_PEAK_GBS,
                            "tuned_kernel": bool(w.plan.tuned)}
            w.free()
            torch.cuda.empty_cache()
        out["other_workloads_1gpu"] = others
    if world > 1:
        barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
