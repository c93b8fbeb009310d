#!/bin/bash
# Effective shader clock of the headline kernel with and without its memory traffic (SPEECH_ABLATE):
# GRBM_GUI_ACTIVE (summed over the 8 XCDs) / 8 / kernel duration.
set -u
OUT=/root/repo/gpurun_out/clock_probe
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for a in ${ABLATES:-0 3}; do
  export SPEECH_ABLATE=$a
  timeout -s KILL 120 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/ablate$a -- python3 /root/repo/bench.py --steps 20 --warmup 2 --no-extras > $OUT/ablate$a.log 2>&1 || { echo "pass $a failed"; exit 1; }
done
echo done
