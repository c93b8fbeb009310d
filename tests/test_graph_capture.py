"""speech_batch_run under HIP graph capture: a run is kernel launches on the caller's stream and nothing else -- no
allocation, no synchronisation, no host copy -- so a serving loop can record it once and replay it.  Through the C ABI: the
launches of every kind (plp: both of its launches, with the frame energies in the batch's scratch; mn0: the moments pass) are
captured into a graph on a side stream, the graph is replayed on NEW audio written into the same buffer, and the features equal
those of a direct call bit for bit; a replay changes nothing in the device memory in use."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

CASES = [("fbank", dict(B=80)), ("mfcc", {}), ("mfcc", dict(B=40, C=40, hif=7600.0, mn0=1)), ("plp", dict(use_energy=1)), ("spectrogram", {}),
         ("fbank", dict(B=80, sr=48000.0, hif=24000.0)), ("mfcc", dict(dither=1.0))]


@pytest.mark.parametrize("kind,o", CASES, ids=["%s-%s" % (k, "-".join("%s=%s" % kv for kv in sorted(o.items()))) for k, o in CASES])
def test_batch_run_is_capturable_and_replays_bit_for_bit(kind, o):
    import torch
    import signals
    from speech_amd import api
    sr = o.get("sr", 16000.0)
    lens = [int(sr * s) for s in (1.0, 0.26, 2.5, 0.4, 1.7)]
    plan = api.Plan(kind, **o)
    b = api.Batch(plan, lens)
    audio = lambda seed: torch.from_numpy(np.concatenate([signals.utterance(n, seed + i) for i, n in enumerate(lens)])).cuda()
    X = audio(300)
    Y = torch.zeros(b.total_out, device="cuda")
    b.run(X, Y)                                                            # a first call outside the capture (lazy module loading)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        b.run(X, Y, stream=torch.cuda.current_stream().cuda_stream)
    for seed in (400, 500):
        Xn = audio(seed)
        want = torch.zeros_like(Y)
        b.run(Xn, want)
        torch.cuda.synchronize()
        X.copy_(Xn)
        Y.zero_()
        torch.cuda.synchronize()
        free0 = torch.cuda.mem_get_info()[0]
        g.replay()
        torch.cuda.synchronize()
        assert torch.cuda.mem_get_info()[0] == free0                       # a replay allocates nothing
        assert torch.equal(Y, want) and bool(torch.isfinite(Y).all()) and float(Y.abs().max()) > 0.0
    del g
    b.close()
    plan.close()
