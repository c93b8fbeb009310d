"""Where the HIP output is furthest from the oracle on the BASELINE option sets (GPU box): for noise-dominated and tonal audio,
e_go, the computed float32 floor (compare.float32_floor), and WHAT the worst element is -- frame, column, and how far below the
frame's largest value the underlying quantity sits -- so that tests/test_gpu_parity.py can list the cases above the floor with
their measured numbers and cause.

    python tools/strict_probe.py > gpurun_out/strict_probe.txt
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np                                                      # noqa: E402
import oracle_lib as O, np_truth as T, signals                          # noqa: E402,E401
import compare                                                          # noqa: E402
from speech_amd import api                                              # noqa: E402
from test_gpu_parity import BASELINE_SETS, LEGACY, TRUTH, _legacy_kwargs  # noqa: E402


def main():
    out = []
    for secs, flat in ((s_, f_) for s_ in (1, 3, 10) for f_ in (True, False)):
        x = signals.utterance(16000 * secs, 31337, flat=flat)
        for kind, o in BASELINE_SETS:
            got = LEGACY[kind](x, **_legacy_kwargs(o))
            _, ref = O.run(kind, x, **o)
            _, reff = O.run(kind, x, exact=False, **o)
            tr = TRUTH[kind](x.astype(np.float64), single=True, **o)
            tol = 1e-5 if kind == "spectrogram" else 1e-4
            floor = compare.float32_floor(kind, o, ref, reff, tr)
            e_go = compare.max_err(kind, o, got, ref)
            d = np.abs(got.astype(np.float64) - ref)
            w, c = np.unravel_index(np.argmax(d), d.shape)
            n0 = len(compare.PARITY_LOG)
            ok, _, e_gt, e_ot = compare.check_parity(kind, o, got, ref, tr, tol=tol, oracle_fast=reff, label="probe")
            # the log-mel / log-power picture of that frame: how far below the frame's largest bin is the bin behind the worst element?
            if kind in ("fbank", "spectrogram"):
                lvl = float(tr[w, c] - tr[w].max()) * 10.0 / np.log(10.0)
            else:
                fb = T.fbank(x.astype(np.float64), single=True, **{k: v for k, v in o.items() if k in ("B", "hif", "snip")})
                lvl = float(fb[w].min() - fb[w].max()) * 10.0 / np.log(10.0)
            rec = dict(kind=kind, opts=o, flat=flat, secs=secs, e_go=e_go, floor=floor, tol=tol, above=bool(e_go > max(tol, floor)), e_gt=e_gt, e_ot=e_ot,
                       e_ff=compare.max_err(kind, o, reff, ref), rms_ratio=compare.PARITY_LOG[n0][6], clause=compare.PARITY_LOG[n0][2],
                       worst_frame=int(w), worst_col=int(c), level_db_below_frame_peak=lvl,
                       frac_above_tol=float((d > tol).mean()))
            out.append(rec)
            print(json.dumps(rec))
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "strict_probe.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
