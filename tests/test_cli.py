"""Drop-in CLIs (cli/kaldi_cli.cpp): flags, defaults, exit codes and messages of the reference tools
(src/kaldi_mfcc.cpp:118-327, srci/kaldi_fbank.cpp:89-106), .npy in and out."""
import os
import subprocess

import numpy as np
import pytest

import signals

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "cli", "bin")
TOOLS = ["kaldi_spectrogram", "kaldi_fbank", "kaldi_mfcc", "kaldi_plp"]


def run(tool, *args, stdin=None):
    return subprocess.run([os.path.join(BIN, tool), *args], input=stdin, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)


@pytest.fixture(scope="module", autouse=True)
def _built():
    if not all(os.path.exists(os.path.join(BIN, t)) for t in TOOLS):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "cli")])


@pytest.mark.parametrize("tool", TOOLS)
def test_help_prints_usage_and_exits_1(tool):
    r = run(tool, "-h")
    assert r.returncode == 1                                    # src/kaldi_mfcc.cpp:142-147
    out = r.stdout.decode()
    assert out.startswith("Usage: " + tool)
    for flag in ("--srate", "--frame_length", "--frame_shift", "--snip_edges", "--dither", "--zero_dc", "--raw_energy", "--preemph",
                 "--wintype", "--zero_mean", "--ofile"):
        assert flag in out
    assert ("--B=" in out) == (tool != "kaldi_spectrogram")
    assert ("--C=" in out) == (tool in ("kaldi_mfcc", "kaldi_plp"))
    assert ("--amp" in out) == (tool == "kaldi_fbank")


def test_validation_messages_and_codes(tmp_path):
    x = tmp_path / "x.npy"
    np.save(x, np.zeros(4000, np.float32))
    for args, msg in [(("-c", "30"), "C (num ceps) must be <= B (num fbank bins)"), (("-p", "1.5"), "p must be in [0.0 1.0]"),
                      (("-d", "-1"), "d must be nonnegative"), (("-l", "9000"), "hif must be > lof"), (("-b", "0"), "B must be positive"),
                      (("-t", "0"), "frame length must be positive")]:
        r = run("kaldi_mfcc", *args, str(x), "-o", str(tmp_path / "y.npy"))
        assert r.returncode == 1 and msg in r.stderr.decode(), (args, r.stderr)
    r = run("kaldi_mfcc", str(tmp_path / "missing.npy"), "-o", str(tmp_path / "y.npy"))
    assert r.returncode == 1 and "problem opening input file" in r.stderr.decode()
    bad = tmp_path / "bad.npy"
    bad.write_bytes(b"not a tensor")
    r = run("kaldi_mfcc", str(bad), "-o", str(tmp_path / "y.npy"))
    assert r.returncode == 1 and "problem reading header for input file" in r.stderr.decode()
    m = tmp_path / "m.npy"
    np.save(m, np.zeros((4, 5), np.float32))
    r = run("kaldi_mfcc", str(m), "-o", str(tmp_path / "y.npy"))
    assert r.returncode == 1 and "input (X) must be a vector" in r.stderr.decode()
    d = tmp_path / "d.npy"
    np.save(d, np.zeros(4000, np.float64))
    r = run("kaldi_fbank", str(d), "-o", str(tmp_path / "y.npy"))
    assert r.returncode == 1 and "input data type must be in {1}" in r.stderr.decode()     # srci/kaldi_fbank.cpp:6
    r = run("kaldi_spectrogram", "-b", "40", str(x), "-o", str(tmp_path / "y.npy"))         # not a spectrogram flag
    assert r.returncode == 1


@pytest.mark.gpu
def test_cli_matches_library(tmp_path):
    from speech_amd import api
    x = signals.utterance(8000, 42)
    xf = tmp_path / "x.npy"
    np.save(xf, x)
    # kaldi_mfcc -e -z -y -d0: Kaldi-style front end
    r = run("kaldi_mfcc", "-e", "-z", "-y", "-d0", "-b40", "-c20", str(xf), "-o", str(tmp_path / "y.npy"))
    assert r.returncode == 0, r.stderr
    assert b"elapsed time = " in r.stderr                       # src/kaldi_mfcc.cpp:321-322
    y = np.load(tmp_path / "y.npy")
    assert np.array_equal(y, api.kaldi_mfcc(x, B=40, C=20))
    # CLI defaults: snip_edges, zero_dc, raw_energy all off; hif = Nyquist; fbank linear unless -g
    r = run("kaldi_fbank", "-d0", "-g", "-u", "-400", "-x", str(xf), "-o", str(tmp_path / "f.npy"))
    assert r.returncode == 0, r.stderr
    f = np.load(tmp_path / "f.npy")
    assert np.array_equal(f, api.kaldi_fbank(x, snip_edges=0, dc0=0, raw_energy=0, hif=7600.0, lg=1, use_energy=1))
    # stdin -> stdout, spectrogram (no timing line: generated without -t in the reference, Makefile:47)
    r = run("kaldi_spectrogram", "-e", "-z", "-d0", "-", stdin=xf.read_bytes())
    assert r.returncode == 0 and b"elapsed" not in r.stderr
    import io
    s = np.load(io.BytesIO(r.stdout))
    assert np.array_equal(s, api.kaldi_spectrogram(x, raw_energy=0))
    # float64 input and Fortran order: D x W, frames contiguous
    xd = tmp_path / "xd.npy"
    with open(xd, "wb") as fh:                                  # numpy itself never writes fortran_order for a vector: build the header by hand
        np.lib.format.write_array_header_1_0(fh, {"descr": "<f8", "fortran_order": True, "shape": (x.size, 1)})
        fh.write(x.astype(np.float64).tobytes())
    r = run("kaldi_plp", "-e", "-z", "-y", "-d0", str(xd), "-o", str(tmp_path / "p.npy"))
    assert r.returncode == 0, r.stderr
    pl = np.load(tmp_path / "p.npy")
    ref = api.kaldi_plp(x.astype(np.float64))
    assert pl.dtype == np.float64 and pl.shape == ref.shape[::-1] and np.isfortran(pl) and np.array_equal(pl.T, ref)
    # default dither 0.1 is on: seeded from the clock as in the reference (c/kaldi_fbank.c:97-101), so two runs differ;
    # --seed pins the noise (and --utt the utterance number), and then two runs agree bit for bit
    a = run("kaldi_mfcc", "-e", str(xf), "-o", str(tmp_path / "a.npy")); b = run("kaldi_mfcc", "-e", str(xf), "-o", str(tmp_path / "b.npy"))
    assert a.returncode == 0 and b.returncode == 0
    ya, yb = np.load(tmp_path / "a.npy"), np.load(tmp_path / "b.npy")
    assert ya.shape == yb.shape and not np.array_equal(ya, yb) and np.abs(ya - yb).max() < 2.0
    outs = []
    for name, extra in (("c", ("--seed=7",)), ("d", ("--seed=7",)), ("e", ("--seed=7", "--utt=3")), ("f", ("--seed=8",))):
        r = run("kaldi_mfcc", "-e", *extra, str(xf), "-o", str(tmp_path / (name + ".npy")))
        assert r.returncode == 0, r.stderr
        outs.append(np.load(tmp_path / (name + ".npy")))
    assert np.array_equal(outs[0], outs[1]) and not np.array_equal(outs[0], outs[2]) and not np.array_equal(outs[0], outs[3])


def read_ark(path):
    from speech_amd import ark
    return {k: (off, m) for k, off, m in ark.read_ark(path)}


def test_ark_module_round_trip(tmp_path):
    from speech_amd import ark
    rng = np.random.default_rng(3)
    feats = {"a": rng.standard_normal((7, 13)).astype(np.float32), "b-2": rng.standard_normal((0, 5)).astype(np.float32),
             "c": rng.standard_normal((3, 4))}
    p, scp = str(tmp_path / "f.ark"), str(tmp_path / "f.scp")
    offs = ark.write_ark(p, feats, scp=scp)
    more = ark.write_ark(p, {"d": feats["a"][:2]}, scp=scp, append=True)
    got = list(ark.read_ark(p))
    assert [g[0] for g in got] == ["a", "b-2", "c", "d"]
    for k, off, m in got:
        want = feats["a"][:2] if k == "d" else feats[k]
        assert m.dtype == want.dtype and np.array_equal(m, want) and off == {**offs, **more}[k]
    assert [(k, m.shape) for k, m in ark.read_scp(scp)] == [("a", (7, 13)), ("b-2", (0, 5)), ("c", (3, 4)), ("d", (2, 13))]
    with pytest.raises(ValueError):
        ark.write_ark(p, {"bad key": feats["a"]})
    open(tmp_path / "t.ark", "wb").write(open(p, "rb").read()[:40])
    with pytest.raises(ValueError):
        list(ark.read_ark(tmp_path / "t.ark"))


def test_ark_flags_validate_on_cpu(tmp_path):
    x = tmp_path / "x.npy"
    np.save(x, np.zeros(4000, np.float32))
    r = run("kaldi_mfcc", "--append", str(x), "-o", str(tmp_path / "y.ark"))
    assert r.returncode == 1 and "--append and --scp need --ark" in r.stderr.decode()
    r = run("kaldi_mfcc", "--ark=a b", str(x), "-o", str(tmp_path / "y.ark"))
    assert r.returncode == 1 and "white space" in r.stderr.decode()
    assert "--ark=<utt-id>" in run("kaldi_fbank", "-h").stdout.decode()


@pytest.mark.gpu
def test_ark_and_scp_output(tmp_path):
    from speech_amd import api
    ark, scp = tmp_path / "feats.ark", tmp_path / "feats.scp"
    xs = {}
    for i, key in enumerate(("utt-a", "utt-b", "utt-c")):
        x = signals.utterance(6000 + 1000 * i, 70 + i)
        xs[key] = x
        np.save(tmp_path / "x.npy", x)
        r = run("kaldi_mfcc", "-e", "-z", "-y", "-d0", "--ark=" + key, "--scp=" + str(scp), *(["--append"] if i else []),
                str(tmp_path / "x.npy"), "-o", str(ark))
        assert r.returncode == 0, r.stderr
    got = read_ark(ark)
    lines = [l.split() for l in open(scp).read().splitlines()]
    assert [l[0] for l in lines] == ["utt-a", "utt-b", "utt-c"]
    for key, loc in lines:
        path, off = loc.rsplit(":", 1)
        assert path == str(ark) and int(off) == got[key][0]
        assert np.array_equal(got[key][1], api.kaldi_mfcc(xs[key]))
    np.save(tmp_path / "xd.npy", xs["utt-a"].astype(np.float64))
    r = run("kaldi_spectrogram", "-e", "-z", "-y", "-d0", "--ark=d", str(tmp_path / "xd.npy"), "-o", str(tmp_path / "d.ark"))
    assert r.returncode == 0
    d = read_ark(tmp_path / "d.ark")["d"][1]
    assert d.dtype == np.float64 and np.array_equal(d, api.kaldi_spectrogram(xs["utt-a"].astype(np.float64)))


def _wav_bytes(x, rate=16000, channels=1, extra_chunk=False, data_size=None, bits=16, tag=1):
    """RIFF/WAVE writer for the tests (stdlib `wave` cannot write LIST chunks or open-ended data sizes)"""
    import struct
    x = np.asarray(x, dtype=np.int16)
    data = x.tobytes()
    fmt = struct.pack("<HHIIHH", tag, channels, rate, rate * channels * bits // 8, channels * bits // 8, bits)
    body = b"WAVE" + b"fmt " + struct.pack("<I", len(fmt)) + fmt
    if extra_chunk:
        body += b"LIST" + struct.pack("<I", 5) + b"abcde" + b"\0"      # odd size: padded to even
    body += b"data" + struct.pack("<I", len(data) if data_size is None else data_size) + data
    return b"RIFF" + struct.pack("<I", len(body)) + body


def test_wav_input_validates_on_cpu(tmp_path):
    """errors that are found before anything touches the GPU"""
    x = (signals.utterance(4000, 3)).astype(np.int16)
    cases = [(_wav_bytes(x, bits=8), (), "only 16-bit PCM is supported"), (_wav_bytes(x, tag=3), (), "only 16-bit PCM is supported"),
             (_wav_bytes(x, rate=8000), ("-r", "16000"), "does not match the .wav header (8000 Hz)"),
             (_wav_bytes(x), ("--channel=1",), "channel not in file"), (b"RIFF\0\0\0\0WAVX", (), "not a RIFF/WAVE file"),
             (_wav_bytes(x)[:30], (), "bad fmt chunk"), (_wav_bytes(x)[:36], (), "no data chunk"), (_wav_bytes(x[:0]), (), "input (X) found to be empty")]
    for i, (blob, args, msg) in enumerate(cases):
        f = tmp_path / ("bad%d.wav" % i)
        f.write_bytes(blob)
        r = run("kaldi_mfcc", "-d0", *args, str(f), "-o", str(tmp_path / "y.npy"))
        assert r.returncode == 1 and msg in r.stderr.decode(), (i, r.stderr)
    r = run("kaldi_mfcc", "-h")
    assert "--channel=<n>" in r.stdout.decode()


@pytest.mark.gpu
def test_wav_input_matches_float_input(tmp_path):
    """a 16-bit PCM .wav goes to the device as int16 and gives the features of the same samples as floats, bit for bit"""
    import io
    from speech_amd import api
    x = signals.utterance(12000, 77).astype(np.int16)
    xf = x.astype(np.float32)
    w = tmp_path / "x.wav"
    w.write_bytes(_wav_bytes(x, extra_chunk=True))
    r = run("kaldi_mfcc", "-e", "-z", "-y", "-d0", str(w), "-o", str(tmp_path / "y.npy"))
    assert r.returncode == 0, r.stderr
    assert np.array_equal(np.load(tmp_path / "y.npy"), api.kaldi_mfcc(xf))
    # sample rate from the header (8 kHz -> 200-sample frames); stdin; open-ended data chunk as streamed files have it
    r = run("kaldi_fbank", "-e", "-z", "-y", "-d0", "-g", "-b", "40", "-", stdin=_wav_bytes(x, rate=8000, data_size=0xFFFFFFFF))
    assert r.returncode == 0, r.stderr
    assert np.array_equal(np.load(io.BytesIO(r.stdout)), api.kaldi_fbank(xf, sr=8000.0, hif=4000.0, B=40))
    # stereo: the chosen channel; a data size beyond the file's end gives what the file holds
    st = np.stack([x, x[::-1]], axis=1)
    w2 = tmp_path / "st.wav"
    w2.write_bytes(_wav_bytes(st, channels=2, data_size=4 * x.size + 400))
    for ch, want in ((0, xf), (1, xf[::-1].copy())):
        r = run("kaldi_spectrogram", "-e", "-z", "-d0", "--channel=%d" % ch, str(w2), "-o", str(tmp_path / "s.npy"))
        assert r.returncode == 0, r.stderr
        assert np.array_equal(np.load(tmp_path / "s.npy"), api.kaldi_spectrogram(want, raw_energy=0))
    # ark output from a .wav, the compute-mfcc-feats use
    r = run("kaldi_plp", "-e", "-z", "-y", "-d0", "--ark=utt1", str(w), "-o", str(tmp_path / "p.ark"))
    assert r.returncode == 0, r.stderr
    from speech_amd import ark
    (key, _, m), = list(ark.read_ark(str(tmp_path / "p.ark")))
    assert key == "utt1" and np.array_equal(m, api.kaldi_plp(xf))


def test_wav_scp_validates_on_cpu(tmp_path):
    x = signals.utterance(4000, 3).astype(np.int16)
    w = tmp_path / "a.wav"
    w.write_bytes(_wav_bytes(x))
    lst = tmp_path / "wav.scp"
    cases = [("utt1 %s\n" % w, ("-o", "-"), "--wav_scp needs an output archive"), ("utt1 %s\n" % w, (), "--wav_scp needs an output archive"),
             ("utt1 %s\n" % w, ("--ark=k", "-o", str(tmp_path / "o.ark")), "takes the utterance ids from the list"),
             ("utt1 sox %s -t wav - |\n" % w, ("-o", str(tmp_path / "o.ark")), "pipes are not supported"),
             ("utt1\n", ("-o", str(tmp_path / "o.ark")), "line without a path"), ("\n\n", ("-o", str(tmp_path / "o.ark")), "lists no utterances"),
             ("utt1 %s\n" % (tmp_path / "missing.wav"), ("-o", str(tmp_path / "o.ark")), "problem opening input file"),
             ("utt1 %s\n" % w, ("-r", "8000", "-o", str(tmp_path / "o.ark")), "does not match the .wav header (16000 Hz)")]
    for i, (text, args, msg) in enumerate(cases):
        lst.write_text(text)
        r = run("kaldi_mfcc", "-d0", "--wav_scp=%s" % lst, *args)
        assert r.returncode == 1 and msg in r.stderr.decode(), (i, r.stderr)
    r = run("kaldi_mfcc", "-d0", "--wav_scp=%s" % lst, str(w), "-o", str(tmp_path / "o.ark"))
    assert r.returncode == 1 and "--wav_scp replaces the input file" in r.stderr.decode()
    r = run("kaldi_mfcc", "-d0", "--wav_scp=%s" % (tmp_path / "nolist"), "-o", str(tmp_path / "o.ark"))
    assert r.returncode == 1 and "problem opening the wav_scp file" in r.stderr.decode()
    # --devices: a comma-separated list of GPU numbers, with --wav_scp only
    for bad in ("--devices=", "--devices=0,,1", "--devices=a", "--devices=0,-1"):
        r = run("kaldi_mfcc", "-d0", bad, "--wav_scp=%s" % lst, "-o", str(tmp_path / "o.ark"))
        assert r.returncode == 1 and "invalid command line" in r.stderr.decode(), bad
    r = run("kaldi_mfcc", "-d0", "--devices=0,1", str(w), "-o", str(tmp_path / "o.npy"))
    assert r.returncode == 1 and "--devices goes with --wav_scp" in r.stderr.decode()


@pytest.mark.gpu
def test_wav_scp_to_archive(tmp_path):
    """the compute-mfcc-feats use: a wav.scp in, one archive + scp out, every entry equal to the single-file run"""
    from speech_amd import api, ark
    lens = [12000, 400, 30000, 8001]
    xs = [signals.utterance(n, 90 + i).astype(np.int16) for i, n in enumerate(lens)]
    lines = []
    for i, x in enumerate(xs):
        f = tmp_path / ("u%d.wav" % i)
        f.write_bytes(_wav_bytes(x, extra_chunk=bool(i & 1)))
        lines.append("utt%d   %s  " % (i, f))
    lst = tmp_path / "wav.scp"
    lst.write_text("\n".join(lines) + "\n\n")
    out, scp = tmp_path / "feats.ark", tmp_path / "feats.scp"
    r = run("kaldi_mfcc", "-e", "-z", "-y", "-d0", "-b", "40", "-c", "20", "--wav_scp=%s" % lst, "-o", str(out), "--scp=%s" % scp)
    assert r.returncode == 0, r.stderr
    assert b"elapsed time = " in r.stderr
    got = list(ark.read_ark(str(out)))
    assert [k for k, _, _ in got] == ["utt%d" % i for i in range(4)]
    for (k, off, m), x in zip(got, xs):
        assert np.array_equal(m, api.kaldi_mfcc(x.astype(np.float32), B=40, C=20)), k
    by_scp = list(ark.read_scp(str(scp)))                     # follows "<utt-id> <ark>:<offset>"
    assert [k for k, _ in by_scp] == [k for k, _, _ in got] and all(np.array_equal(a[1], b[2]) for a, b in zip(by_scp, got))
    # --append adds to the archive with correct offsets; default hif = Nyquist of the files' rate; dither (default 0.1) reproducible under --seed
    r = run("kaldi_fbank", "-g", "--seed=11", "--wav_scp=%s" % lst, "-o", str(tmp_path / "f.ark"))
    r2 = run("kaldi_fbank", "-g", "--seed=11", "--wav_scp=%s" % lst, "-o", str(tmp_path / "f2.ark"))
    r3 = run("kaldi_fbank", "-g", "--wav_scp=%s" % lst, "-o", str(tmp_path / "f3.ark"))
    assert r.returncode == 0 and r2.returncode == 0 and r3.returncode == 0, r.stderr
    assert (tmp_path / "f.ark").read_bytes() == (tmp_path / "f2.ark").read_bytes() != (tmp_path / "f3.ark").read_bytes()
    # --devices=0,0: two pipelines inside the library (speech_run_host_multi); the same archive to the byte, dither included
    r4 = run("kaldi_fbank", "-g", "--seed=11", "--devices=0,0", "--wav_scp=%s" % lst, "-o", str(tmp_path / "f4.ark"))
    assert r4.returncode == 0, r4.stderr
    assert (tmp_path / "f4.ark").read_bytes() == (tmp_path / "f.ark").read_bytes()
    r = run("kaldi_mfcc", "-e", "-z", "-y", "-d0", "-b", "40", "-c", "20", "--append", "--wav_scp=%s" % lst, "-o", str(out), "--scp=%s" % scp)
    assert r.returncode == 0, r.stderr
    again = list(ark.read_ark(str(out)))
    assert len(again) == 8 and all(np.array_equal(again[i][2], again[i + 4][2]) for i in range(4))
    by_scp = list(ark.read_scp(str(scp)))
    assert len(by_scp) == 8 and all(np.array_equal(a[1], b[2]) for a, b in zip(by_scp, again))


def test_raw_input_validates_on_cpu(tmp_path):
    f = tmp_path / "x.raw"
    f.write_bytes(b"\0" * 4001)
    r = run("kaldi_mfcc", "-d0", "--raw=f32", str(f), "-o", str(tmp_path / "y.raw"))
    assert r.returncode == 1 and "not a multiple of the sample size" in r.stderr.decode()
    r = run("kaldi_mfcc", "-d0", "--raw=u8", str(f), "-o", str(tmp_path / "y.raw"))
    assert r.returncode == 1 and "invalid command line" in r.stderr.decode()
    f.write_bytes(b"")
    r = run("kaldi_mfcc", "-d0", "--raw=f32", str(f), "-o", str(tmp_path / "y.raw"))
    assert r.returncode == 1 and "input (X) found to be empty" in r.stderr.decode()
    f.write_bytes(b"\0" * 4000)
    r = run("kaldi_fbank", "-d0", "--raw=f64", str(f), "-o", str(tmp_path / "y.raw"))
    assert r.returncode == 1 and "input data type must be in {1}" in r.stderr.decode()
    r = run("kaldi_mfcc", "-d0", "--raw=f32", "--wav_scp=%s" % f, "-o", str(tmp_path / "y.raw"))
    assert r.returncode == 1 and "exclude each other" in r.stderr.decode()


@pytest.mark.gpu
def test_raw_input_and_output(tmp_path):
    """headerless samples in, headerless frame-major features out, for the three sample types; stdin to stdout too"""
    from speech_amd import api
    x = signals.utterance(9000, 55)
    want = api.kaldi_mfcc(x)
    for typ, data, dt in (("f32", x.astype(np.float32), np.float32), ("f64", x.astype(np.float64), np.float64), ("s16", x.astype(np.int16), np.float32)):
        f = tmp_path / ("x." + typ)
        f.write_bytes(data.tobytes())
        r = run("kaldi_mfcc", "-e", "-z", "-y", "-d0", "--raw=" + typ, str(f), "-o", str(tmp_path / "y.raw"))
        assert r.returncode == 0, r.stderr
        y = np.fromfile(tmp_path / "y.raw", dtype=dt).reshape(-1, 13)
        ref = api.kaldi_mfcc(x.astype(np.float64)) if typ == "f64" else want
        assert np.array_equal(y, ref), typ
    r = run("kaldi_spectrogram", "-e", "-z", "-d0", "-r", "8000", "--raw=s16", "-", stdin=x.astype(np.int16).tobytes())
    assert r.returncode == 0, r.stderr
    s = np.frombuffer(r.stdout, dtype=np.float32).reshape(-1, 129)
    assert np.array_equal(s, api.kaldi_spectrogram(x, sr=8000.0, raw_energy=0))
    # raw in, archive out
    r = run("kaldi_mfcc", "-e", "-z", "-y", "-d0", "--raw=f32", "--ark=u", str(tmp_path / "x.f32"), "-o", str(tmp_path / "y.ark"))
    assert r.returncode == 0, r.stderr
    from speech_amd import ark
    (k, _, m), = list(ark.read_ark(str(tmp_path / "y.ark")))
    assert k == "u" and np.array_equal(m, want)
