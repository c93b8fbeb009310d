"""The frame mean's share of the float32 noise (CPU only).

The reference sums a frame's samples from the last to the first in float32 and divides by L (c/kaldi_mfcc.c:265-271); the
tuned kernels sum each lane's sample pairs over the rows, add the two halves, reduce over the frame's 16 lanes with a DPP tree
(quad_perm [1,0,3,2], [2,3,0,1], row_half_mirror, row_mirror: fast_common.hpp row16_sum) and multiply by float32(1/L).  Both
orders are restated here in numpy float32, bit for bit, and fed as THE ONLY float32 step into the float64 statement of the
pipeline (tests/np_truth.py): the distance of the result from the all-float64 features is what that mean alone contributes,
in the unit of the tolerance.  Beside it: the whole float32 CPU arithmetic's distance (the oracle's), for scale.

    python tools/mean_noise.py > profiles/r3_mean_noise.txt
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np                                                      # noqa: E402
import np_truth as T, oracle_lib as O, signals, compare                  # noqa: E402,E401

f32 = np.float32


def mean_reference(fr):
    """c/kaldi_mfcc.c:265-271: running float32 sum from the last sample to the first, one IEEE division by L"""
    x = fr.astype(f32)
    acc = np.zeros(x.shape[0], dtype=f32)
    for l in range(x.shape[1] - 1, -1, -1):
        acc = (acc + x[:, l]).astype(f32)
    return (acc / f32(x.shape[1])).astype(np.float64)[:, None]


def mean_tuned(fr):
    """fast_common.hpp frame_power: lane n1 (of 16) holds the pairs (x[2 n1 + 32 n2], x[2 n1 + 32 n2 + 1]); packed running sums over
    n2, the two halves added, row16_sum's four DPP steps, times float32(1 / L)"""
    x = fr.astype(f32)
    W, L = x.shape
    nz = (L + 31) // 32
    xp = np.zeros((W, 32 * nz), dtype=f32)
    xp[:, :L] = x
    xp = xp.reshape(W, nz, 16, 2)                                       # [frame][n2][n1][even / odd]
    sx = np.zeros((W, 16), dtype=f32); sy = np.zeros((W, 16), dtype=f32)
    for n2 in range(nz):
        sx = (sx + xp[:, n2, :, 0]).astype(f32)
        sy = (sy + xp[:, n2, :, 1]).astype(f32)
    v = (sx + sy).astype(f32)
    lane = np.arange(16)
    for perm in (lane ^ 1, lane ^ 2, (lane & 8) | (7 - (lane & 7)), 15 - lane):
        v = (v + v[:, perm]).astype(f32)
    inv = f32(1.0) / f32(L)
    return (v[:, :1] * inv).astype(f32).astype(np.float64)


CASES = [("mfcc 23/13", "mfcc", {}), ("fbank-80", "fbank", dict(B=80)), ("mfcc hires 40/40", "mfcc", dict(B=40, C=40, hif=7600.0)),
         ("spectrogram", "spectrogram", {}), ("plp 23/13", "plp", {}), ("mfcc, 17 ms frames", "mfcc", dict(fl=17.0))]
TRUTH = {"spectrogram": T.spectrogram, "fbank": T.fbank, "mfcc": T.mfcc, "plp": T.plp}


def main():
    print("| case | signal | reference-order mean: max / rms | tuned-kernel mean: max / rms | whole float32 CPU arithmetic: max / rms | tuned mean's share of the CPU noise power |")
    print("|---|---|---|---|---|---|")
    for sig, flat, dc in (("noise-dominated", True, 0.0), ("tonal", False, 0.0), ("noise-dominated + DC offset 2000", True, 2000.0)):
        x = (signals.utterance(48000, 31337, flat=flat) + f32(dc)).astype(f32)
        for label, kind, o in CASES:
            T.MEAN_HOOK = None
            tr = TRUTH[kind](x.astype(np.float64), single=True, **o)
            res = []
            for hook in (mean_reference, mean_tuned):
                T.MEAN_HOOK = hook
                y = TRUTH[kind](x.astype(np.float64), single=True, **o)
                res.append((compare.max_err(kind, o, y, tr), compare.rms_err(kind, o, y, tr)))
            T.MEAN_HOOK = None
            _, ref = O.run(kind, x, **o)
            cpu = (compare.max_err(kind, o, ref, tr), compare.rms_err(kind, o, ref, tr))
            print("| %s | %s | %.2e / %.2e | %.2e / %.2e | %.2e / %.2e | %.4f %% |"
                  % (label, sig, res[0][0], res[0][1], res[1][0], res[1][1], cpu[0], cpu[1], 100.0 * (res[1][1] / cpu[1]) ** 2))


if __name__ == "__main__":
    main()
