#!/bin/bash
# cycles, clock and duration of a mel-kind launch as a function of its SIZE: tools/clock_size.sh workload "<utterance counts>"
# (why do 0.5 ms launches cost 20 % more per utterance than 6 ms ones?)
W=$1; AS=$2
cd /tmp && export TMPDIR=/tmp
export SPEECH_AMD_LIB=${LIB:-/root/repo/speech_amd/lib/libspeech.so} SPEECH_PF_DIST=${PF:-0}
for a in $AS; do
  unset SPEECH_ABLATE
  O=/root/repo/gpurun_out/clk_$a
  rm -rf $O
  timeout -s KILL 120 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY --kernel-trace --output-format csv -d $O -- python3 /root/repo/bench.py --steps 20 --warmup 2 --no-extras --workload $W --nutt $a > $O.log 2>&1 || { echo "pass $a failed"; tail -5 $O.log; exit 1; }
  python3 - "$O" "$a" <<'P'
import sys,glob,csv,collections
root,a=sys.argv[1:3]
acc=collections.defaultdict(list)
for path in glob.glob(root+"/**/*counter_collection.csv",recursive=True):
    per=collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(path)):
        if "fast_mel_kernel" in r["Kernel_Name"]: per[r["Dispatch_Id"]][r["Counter_Name"]]+=float(r["Counter_Value"])
    for dd in per.values():
        for k,v in dd.items(): acc[k].append(v)
dur=[]
for path in glob.glob(root+"/**/*kernel_trace.csv",recursive=True):
    for r in csv.DictReader(open(path)):
        if "fast_mel_kernel" in r["Kernel_Name"]: dur.append(float(r["End_Timestamp"])-float(r["Start_Timestamp"]))
m=lambda v: sum(v)/len(v)
cyc=m(acc["GRBM_GUI_ACTIVE"])/8
print("utterances",a,"launches",len(dur),"duration_us %.1f"%(m(dur)/1e3),"cycles %.0f"%cyc,"clock_GHz %.3f"%(cyc/m(dur)),"valu_busy %.3f"%(m(acc["SQ_ACTIVE_INST_VALU"])*4/1024/cyc),"wait_frac %.3f"%(m(acc["SQ_WAIT_ANY"])/m(acc["SQ_WAVE_CYCLES"])))
P
  rm -rf $O
done
