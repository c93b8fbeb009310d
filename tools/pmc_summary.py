"""Summarise rocprofv3 --pmc counter_collection.csv files: mean counter value per launch of the named kernel."""
import csv, glob, sys, collections
root, kern = sys.argv[1], sys.argv[2]
for path in sorted(glob.glob(root + "/pmc_*/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(list)
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(path)):
        if kern in r["Kernel_Name"]:
            per[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
    for d in per.values():
        for k, v in d.items():
            acc[k].append(v)
    for k, v in acc.items():
        print("%-28s launches=%d mean=%.6g min=%.6g max=%.6g" % (k, len(v), sum(v) / len(v), min(v), max(v)))
