"""Regenerates tests/golden/oracle_golden.npz from the CPU oracle (exact-FFT mode).

PARITY UNPINNED: the reference ships no golden vectors and cannot be built in this image (see
oracle/kaldi_oracle.c), so these fixtures pin the ORACLE's own output (a regression anchor that also
travels to the GPU box), not the reference's.  Inputs are regenerated from the seed by tests/signals.py.
Run:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_lib as O  # noqa: E402
import signals  # noqa: E402

CASES = [
    # name, kind, N, seed, single, options
    ("mfcc_cfg1", "mfcc", 4000, 11, True, dict(snip=1)),
    ("mfcc_cfg1_nosnip", "mfcc", 4000, 11, True, dict(snip=0)),
    ("mfcc_hires", "mfcc", 4000, 12, True, dict(B=40, C=40, hif=7600.0)),
    ("mfcc_energy", "mfcc", 3000, 13, True, dict(use_energy=1, raw_energy=0, win="hamming")),
    ("mfcc_d", "mfcc", 3000, 13, False, dict(use_energy=1)),
    ("spec_cfg2", "spectrogram", 2400, 21, True, dict()),
    ("spec_nosnip_hann", "spectrogram", 1700, 22, True, dict(snip=0, win="hann", raw_energy=0, dc0=0)),
    ("spec_d", "spectrogram", 1700, 22, False, dict(snip=0)),
    ("fbank80", "fbank", 4000, 31, True, dict(B=80)),
    ("fbank_amp_energy", "fbank", 3000, 32, True, dict(amp=1, use_energy=1, lg=0, win="blackman", preemph=0.0)),
    ("fbank_8k", "fbank", 3000, 33, True, dict(sr=8000.0, hif=4000.0, B=15, snip=0)),
    ("plp_cfg5", "plp", 4000, 41, True, dict()),
    ("plp_sqrt_energy", "plp", 3000, 42, True, dict(compress=0.5, use_energy=1, snip=0)),
]


def main():
    out = {}
    for name, kind, n, seed, single, opts in CASES:
        sr = opts.get("sr", 16000.0)
        x = signals.utterance(n, seed, sr=sr, dtype=np.float32 if single else np.float64)
        rc, y = O.run(kind, x, single=single, exact=True, **opts)
        assert rc == 0, name
        out[name] = y
    np.savez_compressed(os.path.join(HERE, "oracle_golden.npz"), **out)
    print("wrote", len(out), "cases,", sum(v.nbytes for v in out.values()), "bytes raw")


if __name__ == "__main__":
    main()
