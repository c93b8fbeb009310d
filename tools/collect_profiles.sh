#!/bin/bash
# copies what tools/gpu_prof.sh left under gpurun_out/ (scratch) into profiles/ (tracked): tools/collect_profiles.sh <round tag, e.g. r3>
set -eu
cd "$(dirname "$0")/.."
R=${1:-r3}
for d in gpurun_out/prof_${R}_*; do
  wl=${d#gpurun_out/prof_${R}_}
  cp "$d/kernel_stats.csv" "profiles/${R}_${wl}_kernel_stats.csv"
  { cat "$d/pmc_summary.txt"; echo "# command: $(cat "$d/command.txt")"; } > "profiles/${R}_${wl}_pmc.txt"
done
cp gpurun_out/${R}_counters.json profiles/${R}_counters.json
[ -f gpurun_out/${R}_bench.json ] && cp gpurun_out/${R}_bench.json profiles/${R}_bench.json
[ -f gpurun_out/${R}_parity_clauses.json ] && cp gpurun_out/${R}_parity_clauses.json profiles/${R}_parity_clauses_gpu_suite.json
ls profiles/${R}_*
