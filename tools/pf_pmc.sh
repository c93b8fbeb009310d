#!/bin/bash
# does the mel kernels' L2 prefetch land in the right L2?  FETCH_SIZE and the L2 hit / miss counts with the prefetch off and on
# (libspeech_diag.so, SPEECH_PF_DIST): tools/pf_pmc.sh workload "<distances>"
W=$1; DS=$2
cd /tmp && export TMPDIR=/tmp
export SPEECH_AMD_LIB=/root/repo/speech_amd/lib/libspeech_diag.so
for d in $DS; do
  export SPEECH_PF_DIST=$d
  for c in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"; do
    O=/root/repo/gpurun_out/pfpmc_${d}_${c%% *}
    rm -rf $O
    timeout -s KILL 120 rocprofv3 --pmc $c --output-format csv -d $O -- python3 /root/repo/bench.py --steps 5 --warmup 1 --no-extras --no-check --workload $W > $O.log 2>&1 || { echo "pmc $c failed"; tail -5 $O.log; exit 1; }
    python3 - "$O" "$d" <<'P'
import sys,glob,csv,collections
root,d=sys.argv[1:3]
acc=collections.defaultdict(list)
for path in glob.glob(root+"/**/*counter_collection.csv",recursive=True):
    per=collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(path)):
        if "fast_mel_kernel" in r["Kernel_Name"]: per[r["Dispatch_Id"]][r["Counter_Name"]]+=float(r["Counter_Value"])
    for dd in per.values():
        for k,v in dd.items(): acc[k].append(v)
for k,v in acc.items(): print("pf_dist",d,k,"mean per launch %.6g over %d launches"%(sum(v)/len(v),len(v)))
P
    rm -rf $O
  done
done
