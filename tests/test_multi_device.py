"""speech_run_host_multi (include/speech_hip.h): the in-library runner over several devices -- LPT sharding in C, a host
thread + plan + two-stream staging pipeline per device, no collective (SURVEY.md 8(b) extension row, 8(e)).  The test box has
ONE GPU: devices = [0, 0] runs two pipelines on it, which exercises everything but the second card."""
import numpy as np
import pytest

import signals
from speech_amd import api

pytestmark = pytest.mark.gpu


def _corpus(n=21, sr=16000.0, seed=1700):
    rng = np.random.default_rng(seed)
    lens = [int(v) for v in rng.integers(400, 40000, size=n)] + [400, 160000, 401]
    return [signals.utterance(m, seed + i, sr=sr) for i, m in enumerate(lens)]


@pytest.mark.parametrize("kind,o,dtype,pcm", [("mfcc", dict(dither=1.0), "f32", False), ("fbank", dict(B=80), "f32", True),
                                               ("spectrogram", dict(snip=0, dither=0.5), "f32", False), ("plp", dict(mn0=1), "f64", False)])
def test_two_pipelines_equal_the_single_device_run(kind, o, dtype, pcm):
    """devices = {0, 0}: bit for bit the features of Plan.run_host on one device, dither on (the noise is keyed by the
    utterance's number in the caller's list, not by the shard), for float, double and 16-bit PCM input, any chunk size"""
    xs = _corpus()
    t = np.float32 if dtype == "f32" else np.float64
    host = [x.astype(np.int16) if pcm else x.astype(t) for x in xs]
    plan = api.Plan(kind, dtype=dtype, **o)
    want = plan.run_host(host)
    for devices, chunk in (([0, 0], 0), ([0, 0, 0], 150000), ([0], 0)):
        got = api.run_host_multi(kind, host, devices, dtype=dtype, chunk_bytes=chunk, **o)
        assert len(got) == len(want)
        for u, (g, w) in enumerate(zip(got, want)):
            assert g.dtype == t and g.shape == w.shape and np.array_equal(g, w), (devices, u)
    # explicit corpus numbers: a caller that holds a slice of a larger list
    ids = np.arange(len(xs), dtype=np.uint32)[::-1].copy()
    a = api.run_host_multi(kind, host, [0, 0], dtype=dtype, ids=ids, **o)
    b = plan.run_host(host, ids=ids)
    for g, w in zip(a, b):
        assert np.array_equal(g, w)


def test_a_failed_device_is_requeued_on_the_working_one(capfd):
    """per-device failure handling: device 99 does not exist, so its share fails at plan creation and is
    dealt again to device 0; the call succeeds, says so on stderr, and the features are those of the single-device run"""
    xs = _corpus(9)
    plan = api.Plan("mfcc")
    want = plan.run_host(xs)
    got = api.run_host_multi("mfcc", xs, [0, 99])
    err = capfd.readouterr().err
    assert "1 device(s) failed" in err and "go to the 1 working device(s)" in err
    for g, w in zip(got, want):
        assert np.array_equal(g, w)
    with pytest.raises(api.SpeechError) as e:
        api.run_host_multi("mfcc", xs, [98, 99])
    assert "every device failed" in str(e.value)
    # and the library still works afterwards
    assert np.array_equal(api.run_host_multi("mfcc", xs[:3], [0, 0])[1], want[1])


def test_more_devices_than_utterances_and_invalid_utterances(capfd):
    xs = _corpus(0)                                                          # three utterances
    want = api.Plan("fbank").run_host(xs)
    got = api.run_host_multi("fbank", xs, [0] * 8)
    for g, w in zip(got, want):
        assert np.array_equal(g, w)
    with pytest.raises(api.SpeechError):                                      # rejected before any thread starts
        api.run_host_multi("fbank", xs + [np.zeros(100, dtype=np.float32)], [0, 0])
    assert "frame length must be < signal length if snip_edges" in capfd.readouterr().err
