"""plp's spectral compression: pow_split (exponent / mantissa split, ~1.5 ulp) against exp2(c log2 x) on the whole value (~4 ulp)
-- what the accuracy buys, measured (GPU box).  Runs under the library SPEECH_AMD_LIB names: rms noise ratio and largest
deviation of the default plp option set (compress = 1/3) over 8 signals of 1 s and 2 of 10 s, noise-dominated and tonal."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np                                                      # noqa: E402
import oracle_lib as O, np_truth as T, signals, compare                  # noqa: E402,E401
from speech_amd import api                                              # noqa: E402

rows = []
for flat in (True, False):
    for N, seeds in ((16000, range(8)), (160000, range(2))):
        got, ref, reff, tr = [], [], [], []
        for s in seeds:
            x = signals.utterance(N, 4000 + s, flat=flat)
            got.append(api.kaldi_plp(x)); ref.append(O.run("plp", x)[1]); reff.append(O.run("plp", x, exact=False)[1])
            tr.append(T.plp(x.astype(np.float64), single=True))
        g, r, rf, t = (compare.pool(a) for a in (got, ref, reff, tr))
        r_ot = max(compare.rms_err("plp", {}, r, t), compare.rms_err("plp", {}, rf, t))
        print("%s N=%d x %d: rms(hip-f64) %.3e  rms(f32cpu-f64) %.3e  ratio %.3f  max hip-vs-oracle %.3e  max hip-vs-f64 %.3e  max cpu-vs-f64 %.3e"
              % ("noise" if flat else "tonal", N, len(list(seeds)), compare.rms_err("plp", {}, g, t), r_ot, compare.rms_err("plp", {}, g, t) / r_ot,
                 compare.max_err("plp", {}, g, r), compare.max_err("plp", {}, g, t), max(compare.max_err("plp", {}, r, t), compare.max_err("plp", {}, rf, t))))
