"""speech_batch_cmvn_noalloc (include/speech_hip.h): the cmvn pass from a static device scratch, for callers that cannot
allocate.  Through the C ABI, against speech_batch_cmvn, bit for bit, on two calls in a row, the second with a smaller
input; sizes beyond the scratch are rejected with a message; the call allocates no device memory."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _features(plan, lens, seed):
    import torch
    import signals
    from speech_amd import api
    b = api.Batch(plan, lens)
    X = torch.from_numpy(np.concatenate([signals.utterance(n, seed + i) for i, n in enumerate(lens)])).cuda()
    Y = torch.zeros(b.total_out, device="cuda")
    b.run(X, Y)
    torch.cuda.synchronize()
    return b, Y


@pytest.mark.parametrize("norm_vars", [0, 1])
def test_no_alloc_cmvn_equals_cmvn_twice_in_a_row(norm_vars):
    import torch
    from speech_amd import _lib, api
    lib = _lib.load()
    plan = api.Plan("fbank", B=80)
    for lens, seed in (([16000, 4000, 30000, 401, 12345, 8000, 400], 100), ([5000, 700], 200)):       # the second call is smaller
        b, Y = _features(plan, lens, seed)
        want = Y.clone()
        assert lib.speech_batch_cmvn(b._h, C.c_void_p(want.data_ptr()), norm_vars, None) == 0
        torch.cuda.synchronize()
        got = Y.clone()
        free0 = torch.cuda.mem_get_info()[0]
        assert lib.speech_batch_cmvn_noalloc(b._h, C.c_void_p(got.data_ptr()), norm_vars, None) == 0
        torch.cuda.synchronize()
        assert torch.cuda.mem_get_info()[0] == free0                       # nothing was allocated on the device
        assert torch.equal(got, want) and not torch.equal(got, Y)
        if not norm_vars:                                                  # and it is the mean normalisation: column means vanish
            W0 = int(b.n_frames[0])
            assert float(got[:W0 * 80].view(W0, 80).double().mean(0).abs().max()) < 1e-4
        b.close()


def test_no_alloc_cmvn_rejects_what_the_scratch_cannot_hold(capfd):
    import torch
    from speech_amd import _lib, api
    lib = _lib.load()
    plan = api.Plan("spectrogram")                                         # D = 257: 1021 utterances x 2 x 257 doubles > 4 MiB
    b = api.Batch(plan, [400] * 1021)
    Y = torch.zeros(b.total_out, device="cuda")
    assert lib.speech_batch_cmvn_noalloc(b._h, C.c_void_p(Y.data_ptr()), 0, None) == 1
    assert b"static scratch holds" in lib.speech_last_error()
    b2 = api.Batch(plan, [400] * 1020)                                     # 1020 x 2 x 257 = 524 280 doubles: fits
    Y2 = torch.ones(b2.total_out, device="cuda")
    assert lib.speech_batch_cmvn_noalloc(b2._h, C.c_void_p(Y2.data_ptr()), 0, None) == 0
    torch.cuda.synchronize()
    assert float(Y2.abs().max()) == 0.0                                    # one frame per utterance: the mean is the frame
    capfd.readouterr()


def test_no_alloc_cmvn_double_precision_and_graph_capture():
    import torch
    import signals
    from speech_amd import _lib, api
    lib = _lib.load()
    plan = api.Plan("mfcc", dtype="f64", snip=0)
    lens = [3000, 9000, 500]
    b = api.Batch(plan, lens)
    X = torch.from_numpy(np.concatenate([signals.utterance(n, 300 + i, dtype=np.float64) for i, n in enumerate(lens)])).cuda()
    Y = torch.zeros(b.total_out, device="cuda", dtype=torch.float64)
    b.run(X, Y)
    torch.cuda.synchronize()
    want = Y.clone()
    assert lib.speech_batch_cmvn(b._h, C.c_void_p(want.data_ptr()), 1, None) == 0
    torch.cuda.synchronize()
    # under stream capture, where hipMalloc is not allowed: the call is captured and the replay gives the same features
    Yg = Y.clone()
    graph = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        with torch.cuda.graph(graph, stream=side):
            assert lib.speech_batch_cmvn_noalloc(b._h, C.c_void_p(Yg.data_ptr()), 1, C.c_void_p(torch.cuda.current_stream().cuda_stream)) == 0
    Yg.copy_(Y)
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(Yg, want)
