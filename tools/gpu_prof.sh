#!/bin/bash
# profiles of the round's final kernels: tools/gpu_prof.sh <tag-suffix> workload:kernel-substring ...
set -u
cd /root/repo
SUF=$1; shift
for spec in "$@"; do
  wl=${spec%%:*}; kern=${spec##*:}
  bash tools/profile_round.sh ${SUF}_$wl $wl || exit 1
  python tools/pmc_to_json.py gpurun_out/prof_${SUF}_$wl $wl "$kern" gpurun_out/r3_counters.json gpurun_out/prof_${SUF}_$wl/pmc_summary.txt > /dev/null || exit 1
  cp $(find gpurun_out/prof_${SUF}_$wl/trace -name "*kernel_stats.csv" | head -1) gpurun_out/prof_${SUF}_$wl/kernel_stats.csv
  rm -rf gpurun_out/prof_${SUF}_$wl/trace gpurun_out/prof_${SUF}_$wl/pmc_fetch gpurun_out/prof_${SUF}_$wl/pmc_write gpurun_out/prof_${SUF}_$wl/pmc_sq gpurun_out/prof_${SUF}_$wl/pmc_sq2
done
cat gpurun_out/r3_counters.json | python -c "
import sys,json
for k,v in json.load(sys.stdin).items(): print(k, 'ns', v.get('kernel_avg_ns_under_profiler'), 'valu', round(v.get('valu_util') or 0,3), 'mfma', round(v.get('mfma_util') or 0,3), 'ldsconf', round(v.get('lds_bank_conflict_frac') or 0,3), 'wait', round(v.get('wait_frac') or 0,3), 'stall', round(v.get('issue_stall_frac') or 0,3), 'hbm', v.get('hbm_bytes_per_launch'))"
