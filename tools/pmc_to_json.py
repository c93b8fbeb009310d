"""Turn the rocprofv3 passes of tools/profile_round.sh into the numbers bench.py reports beside the HBM fraction.

    python tools/pmc_to_json.py gpurun_out/prof_<tag> <workload> <kernel substring> profiles/r2_counters.json [summary.txt]

Per launch of the named kernel (mean over the launches of the pass):
  hbm_bytes_per_launch = 2 x FETCH_SIZE + WRITE_SIZE   (KiB -> bytes; gfx950 tallies 128-B read requests at 64 B: MI355X_MICROARCH.md, HBM)
  valu_util            = SQ_ACTIVE_INST_VALU x 4 / SIMDs / cycles      (rocprofiler's VALUBusy; cycles = GRBM_GUI_ACTIVE / 8 XCDs)
  mfma_util            = SQ_VALU_MFMA_BUSY_CYCLES / (cycles x CUs x 4) (rocprofiler's MfmaUtil)
  lds_bank_conflict_frac = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE
  wait_frac / issue_stall_frac = SQ_WAIT_ANY, SQ_WAIT_INST_ANY over SQ_WAVE_CYCLES
"""
import collections
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from speech_amd.srchash import kernel_sources_sha  # noqa: E402

CUS, SIMDS, XCDS = 256, 1024, 8


def means(root, kern):
    out = {}
    for path in sorted(glob.glob(root + "/pmc_*/**/*counter_collection.csv", recursive=True)):
        pas = path.split("/pmc_")[1].split("/")[0]
        per = collections.defaultdict(lambda: collections.defaultdict(float))
        for r in csv.DictReader(open(path)):
            if kern in r["Kernel_Name"]:
                per[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
        acc = collections.defaultdict(list)
        for d in per.values():
            for k, v in d.items():
                acc[k].append(v)
        for k, v in acc.items():
            out[(pas, k)] = (sum(v) / len(v), len(v), min(v), max(v))
    return out


def kernel_time_ns(root, kern):
    for path in glob.glob(root + "/trace/**/*kernel_stats.csv", recursive=True):
        for r in csv.DictReader(open(path)):
            if kern in r["Name"]:
                return float(r["AverageNs"]), int(r["Calls"]), r["Name"]
    return None, 0, None


def main():
    root, workload, kern, dst = sys.argv[1:5]
    m = means(root, kern)
    g = lambda pas, k: m.get((pas, k), (None,))[0]                      # noqa: E731
    avg_ns, calls, name = kernel_time_ns(root, kern)
    rec = {"kernel": name, "kernel_avg_ns_under_profiler": avg_ns, "launches": calls, "source": os.path.basename(root.rstrip("/")),
           "command": open(os.path.join(root, "command.txt")).read().strip() if os.path.exists(os.path.join(root, "command.txt")) else None}
    # what the counters were measured on: the hash of the kernel sources (bench.py nulls the counter fields when the library it
    # runs was built from other sources) and, when the caller passes it, the commit (the GPU box has no .git)
    rec["kernel_sources_sha"] = kernel_sources_sha()
    rec["commit"] = os.environ.get("SPEECH_COMMIT")
    fs, ws = g("fetch", "FETCH_SIZE"), g("write", "WRITE_SIZE")
    if fs is not None and ws is not None:
        rec.update(fetch_size_kib=fs, write_size_kib=ws, hbm_bytes_per_launch=(2.0 * fs + ws) * 1024.0)
    cyc = g("sq", "GRBM_GUI_ACTIVE")
    if cyc:
        cyc /= XCDS
        rec["gpu_cycles_per_launch"] = cyc
        rec["valu_util"] = g("sq", "SQ_ACTIVE_INST_VALU") * 4.0 / SIMDS / cyc
        wc = g("sq", "SQ_WAVE_CYCLES")
        rec["wait_frac"] = g("sq", "SQ_WAIT_ANY") / wc
        rec["issue_stall_frac"] = g("sq", "SQ_WAIT_INST_ANY") / wc
        rec["valu_insts_per_launch"] = g("sq", "SQ_INSTS_VALU")
        rec["shader_clock_ghz"] = cyc / avg_ns if avg_ns else None
    cyc2 = g("sq2", "GRBM_GUI_ACTIVE")
    if cyc2:
        cyc2 /= XCDS
        rec["mfma_util"] = g("sq2", "SQ_VALU_MFMA_BUSY_CYCLES") / (cyc2 * CUS * 4.0)
        rec["mfma_insts_per_launch"] = g("sq2", "SQ_INSTS_MFMA")
        rec["lds_util"] = g("sq2", "SQ_LDS_IDX_ACTIVE") / (cyc2 * CUS)
        rec["salu_insts_per_launch"] = g("sq2", "SQ_INSTS_SALU")
        if g("sq", "SQ_LDS_BANK_CONFLICT") is not None:
            rec["lds_bank_conflict_frac"] = g("sq", "SQ_LDS_BANK_CONFLICT") / g("sq2", "SQ_LDS_IDX_ACTIVE")
    allj = json.load(open(dst)) if os.path.exists(dst) else {}
    allj[workload] = rec
    json.dump(allj, open(dst, "w"), indent=1, sort_keys=True)
    if len(sys.argv) > 5:
        with open(sys.argv[5], "w") as f:
            for (pas, k), (mean, n, lo, hi) in sorted(m.items()):
                f.write("%-8s %-28s launches=%d mean=%.6g min=%.6g max=%.6g\n" % (pas, k, n, mean, lo, hi))
    print(json.dumps(rec, indent=1))


if __name__ == "__main__":
    main()
