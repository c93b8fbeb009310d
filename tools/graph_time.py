"""Launch-bound small batches (a serving loop: one 10 s utterance per call, features stay on the device): K speech_batch_run calls
issued one by one against the same K launches recorded once in a HIP graph and replayed.   python tools/graph_time.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from speech_amd import api

K = 200
for kind, o in (("mfcc", {}), ("fbank", dict(B=80)), ("plp", {})):
    for secs in (1.0, 10.0):
        plan = api.Plan(kind, **o)
        b = api.Batch(plan, [int(16000 * secs)])
        X = torch.randn(b.total_samples, device="cuda") * 1000.0
        Y = torch.zeros(b.total_out, device="cuda")
        s = torch.cuda.current_stream().cuda_stream
        for _ in range(5): b.run(X, Y, stream=s)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(K): b.run(X, Y, stream=s)
        torch.cuda.synchronize()
        direct = (time.perf_counter() - t0) / K * 1e6
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            cs = torch.cuda.current_stream().cuda_stream
            for _ in range(K): b.run(X, Y, stream=cs)
        g.replay(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3): g.replay()
        torch.cuda.synchronize()
        graph = (time.perf_counter() - t0) / (3 * K) * 1e6
        print("%s, one %.0f s utterance per call: %.1f us per call issued one by one, %.1f us in a replayed graph of %d (%.2f M / %.2f M audio-s/s)"
              % (kind, secs, direct, graph, K, secs / direct, secs / graph))
        del g; b.close(); plan.close()
