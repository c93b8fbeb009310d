"""Parity table (GPU box): for every case of tests/cases.py and the BASELINE.json option sets at 1 s and 10 s, the distances
HIP-vs-oracle (e_go), HIP-vs-float64 (e_gt), float32-CPU-vs-float64 (e_ot: the larger of the exact-FFT and the float-FFT
oracle), oracle(float FFT)-vs-oracle(exact FFT) (e_ff), the tolerance, and which clause of tests/compare.check_parity
passed.  Writes <out>.json and <out>.md (committed under profiles/).

    python tools/parity_table.py profiles/r2_parity_table
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np                                                      # noqa: E402
import oracle_lib as O, np_truth as T, signals                          # noqa: E402,E401
import compare                                                          # noqa: E402
from cases import GRID, case_id                                         # noqa: E402
from speech_amd import api                                              # noqa: E402

TRUTH = {"spectrogram": T.spectrogram, "fbank": T.fbank, "mfcc": T.mfcc, "plp": T.plp}
LEGACY = {"spectrogram": api.kaldi_spectrogram, "fbank": api.kaldi_fbank, "mfcc": api.kaldi_mfcc, "plp": api.kaldi_plp}
ALIAS = {"fl": "frame_length", "fs": "frame_shift", "snip": "snip_edges", "win": "win_type"}
BASE = [("config1 mfcc 23/13", "mfcc", {}), ("config2 spectrogram", "spectrogram", {}), ("config3 fbank-80", "fbank", dict(B=80)),
        ("config4 mfcc hires 40/40", "mfcc", dict(B=40, C=40, hif=7600.0)), ("config5 plp 23/13", "plp", {})]


def evaluate(kind, o, x):
    oo = dict(o)
    mn0 = oo.pop("mn0", 0)
    if "order" in o:
        import torch
        plan = api.Plan(kind, **{k: v for k, v in o.items()})
        b = api.Batch(plan, [x.size])
        Y = torch.zeros(b.total_out, device="cuda")
        b.run(torch.from_numpy(x).cuda(), Y); torch.cuda.synchronize()
        got = Y.cpu().numpy().reshape(-1, plan.D)
    else:
        got = LEGACY[kind](x, **{ALIAS.get(k, k): v for k, v in o.items()})
    _, ref = O.run(kind, x, **o)
    _, reff = O.run(kind, x, exact=False, **o)
    tr = TRUTH[kind](x.astype(np.float64), single=True, **oo)
    if mn0:
        tr = T.mean_norm(tr)
    return got, ref, reff, tr


def row(label, kind, o, x, more=None):
    """more: seed offset -> another signal of the same length; cases shorter than compare.POOL_MIN_FRAMES frames are judged on
    compare.POOL_SEEDS signals together (compare.py)"""
    mn0 = o.get("mn0", 0)
    parts = [evaluate(kind, o, x)]
    if more is not None and parts[0][0].shape[0] < compare.POOL_MIN_FRAMES:
        parts += [evaluate(kind, o, more(k)) for k in range(1, compare.POOL_SEEDS)]
        label += " [pooled over %d signals]" % compare.POOL_SEEDS
    got, ref, reff, tr = (compare.pool([p[i] for p in parts]) for i in range(4))
    tol = 1e-5 if (kind == "spectrogram" and not mn0) else 1e-4
    n0 = len(compare.PARITY_LOG)
    ok, e_go, e_gt, e_ot = compare.check_parity(kind, o, got, ref, tr, tol=tol, oracle_fast=reff, label=label)
    clause, ratio = compare.PARITY_LOG[n0][2], compare.PARITY_LOG[n0][6]
    return dict(rms_ratio=ratio, case=label, kind=kind, tol=tol, e_go=e_go, e_gt=e_gt, e_ot=e_ot, e_ff=compare.max_err(kind, o, reff, ref), clause=clause,
                strict_bound=max(tol, compare.float32_floor(kind, o, ref, reff, tr)))


def main():
    out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "parity_table")
    rows = []
    for secs in (1, 10):
        for sig, flat in (("noise-dominated", True), ("tonal", False)):
            x = signals.utterance(16000 * secs, 31337, flat=flat)
            for label, kind, o in BASE:
                rows.append(row("%s, %d s, %s audio" % (label, secs, sig), kind, o, x))
    for c in GRID:
        kind, N, seed, o = c
        rows.append(row(case_id(c), kind, o, signals.utterance(N, seed, sr=o.get("sr", 16000.0), dtype=np.float32),
                        more=lambda k, N=N, seed=seed, o=o: signals.utterance(N, seed + 10007 * k, sr=o.get("sr", 16000.0), dtype=np.float32)))
    n = len(rows)
    n1, n2, n0 = (sum(1 for r in rows if r["clause"] == k) for k in (1, 2, 0))
    summ = dict(cases=n, clause1=n1, clause2=n2, failed=n0, margin=compare.MARGIN, cap_factors=compare.CAP_FACTORS, max_margin=compare.MAX_MARGIN,
                worst_rms_ratio=max((r["rms_ratio"] for r in rows), default=None), worst_ratio_e_gt_over_e_ot=max((r["e_gt"] / r["e_ot"] for r in rows if r["clause"] != 1 and r["e_ot"] > 0), default=None),
                worst_e_go_over_tol_clause2=max((r["e_go"] / r["tol"] for r in rows if r["clause"] == 2), default=None),
                strict_violations=[r["case"] for r in rows if r["e_go"] > r["strict_bound"]])
    json.dump(dict(summary=summ, rows=rows), open(out + ".json", "w"), indent=1)
    with open(out + ".md", "w") as f:
        f.write("# Parity table (tools/parity_table.py, MI355X)\n\n")
        f.write("e_go = HIP vs oracle, e_gt = HIP vs float64 statement, e_ot = float32 CPU (oracle, exact or float FFT, the larger) vs float64, "
                "e_ff = oracle float FFT vs oracle exact FFT; unit: tests/compare.max_err.  Clause 1: e_go <= tol.  Clause 2: "
                "rms(HIP - f64) <= %.2f rms(f32 CPU - f64) [column rms ratio], e_gt <= max(tol, %.1f e_ot) and e_go <= cap (per kind, x tol: %s; plp compress=1: that or %.0f e_ff).  strict bound = max(tol, largest distance among the three CPU evaluations).\n\n" % (compare.MARGIN, compare.MAX_MARGIN, compare.CAP_FACTORS, compare.PLP_C1_CAP_FF))
        f.write("%d cases: %d by clause 1, %d by clause 2, %d failed; cases with e_go above the strict bound: %d\n\n" % (n, n1, n2, n0, len(summ["strict_violations"])))
        kinds = ("spectrogram", "fbank", "mfcc", "plp")
        f.write("| kind | cases | clause 1 | clause 2 | clause-2 share | largest e_go / tol | largest e_gt / e_ot (clause 2) | rms ratio range (clause 2) |\n|---|---|---|---|---|---|---|---|\n")
        for k in kinds:
            rk = [r for r in rows if r["kind"] == k]
            r2 = [r for r in rk if r["clause"] == 2]
            f.write("| %s | %d | %d | %d | %.0f %% | %.1f | %s | %s |\n" % (k, len(rk), sum(1 for r in rk if r["clause"] == 1), len(r2), 100.0 * len(r2) / max(1, len(rk)),
                    max((r["e_go"] / r["tol"] for r in rk if not (k == "plp" and "compress=1.0" in r["case"])), default=0.0),
                    ("%.2f" % max(r["e_gt"] / r["e_ot"] for r in r2)) if r2 else "-", ("%.2f - %.2f" % (min(r["rms_ratio"] for r in r2), max(r["rms_ratio"] for r in r2))) if r2 else "-"))
        f.write("\n")
        f.write("| case | tol | e_go | e_gt | e_ot | e_ff | rms ratio | strict bound | clause |\n|---|---|---|---|---|---|---|---|---|\n")
        for r in rows:
            f.write("| %s | %.0e | %.2e | %.2e | %.2e | %.2e | %.2f | %.2e | %d |\n" % (r["case"], r["tol"], r["e_go"], r["e_gt"], r["e_ot"], r["e_ff"], r["rms_ratio"], r["strict_bound"], r["clause"]))
    print(json.dumps(summ))


if __name__ == "__main__":
    main()
