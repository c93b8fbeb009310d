"""Kaldi binary matrix archives: host-side reader and writer.

The byte layout is the one the reference's kaldi2bin scans for (src/kaldi2bin.cpp:141-175): per entry
``<utt-id> `` ``\\0B`` ``FM `` (``DM `` for double) ``\\4`` int32 rows ``\\4`` int32 cols, then rows x cols values, frame-major.
The CLIs write the same bytes with --ark (cli/kaldi_cli.cpp); nothing here touches the GPU.
"""
import numpy as np

_KINDS = {b"FM ": np.dtype("<f4"), b"DM ": np.dtype("<f8")}


def write_ark(path, feats, scp=None, append=False):
    """feats: mapping or iterable of (utt_id, W x D array of float32/float64).  Returns {utt_id: offset of the \\0B}."""
    items = [(k, np.ascontiguousarray(m)) for k, m in (feats.items() if hasattr(feats, "items") else feats)]
    for key, m in items:                                                    # validate before the file is touched
        if m.ndim != 2 or m.dtype not in (np.float32, np.float64):
            raise ValueError("ark entries must be 2-D float32/float64 matrices")
        if not key or any(c.isspace() for c in key):
            raise ValueError("ark utterance id must be non-empty without white space")
    offsets = {}
    with open(path, "ab" if append else "wb") as f:
        for key, m in items:
            f.write(key.encode() + b" ")
            offsets[key] = f.tell()
            f.write(b"\0B" + (b"FM " if m.dtype == np.float32 else b"DM "))
            f.write(b"\4" + np.int32(m.shape[0]).tobytes() + b"\4" + np.int32(m.shape[1]).tobytes())
            f.write(m.astype(m.dtype.newbyteorder("<"), copy=False).tobytes())
    if scp is not None:
        with open(scp, "a") as s:
            for key, off in offsets.items():
                s.write("%s %s:%d\n" % (key, path, off))
    return offsets


def _read_matrix(f):
    if f.read(2) != b"\0B":
        raise ValueError("not a Kaldi binary entry")
    kind = f.read(3)
    if kind not in _KINDS:
        raise ValueError("unsupported matrix type %r (only FM / DM)" % kind)
    dt = _KINDS[kind]
    hdr = f.read(10)
    if len(hdr) != 10 or hdr[0] != 4 or hdr[5] != 4:
        raise ValueError("malformed matrix header")
    rows, cols = (int(v) for v in np.frombuffer(hdr[1:5] + hdr[6:10], dtype="<i4"))
    if rows < 0 or cols < 0:
        raise ValueError("negative matrix size")
    buf = f.read(rows * cols * dt.itemsize)
    if len(buf) != rows * cols * dt.itemsize:
        raise ValueError("truncated matrix data")
    return np.frombuffer(buf, dtype=dt).reshape(rows, cols)


def read_ark(path):
    """yield (utt_id, offset, matrix) for every entry of a binary archive"""
    with open(path, "rb") as f:
        while True:
            key = bytearray()
            while True:
                c = f.read(1)
                if not c:
                    if key:
                        raise ValueError("truncated archive key")
                    return
                if c == b" ":
                    break
                key += c
            off = f.tell()
            yield key.decode(), off, _read_matrix(f)


def read_scp(path):
    """yield (utt_id, matrix) following '<utt-id> <ark>:<offset>' lines"""
    with open(path) as s:
        for line in s:
            if not line.strip():
                continue
            key, loc = line.split(None, 1)
            ark, off = loc.strip().rsplit(":", 1)
            with open(ark, "rb") as f:
                f.seek(int(off))
                yield key, _read_matrix(f)
