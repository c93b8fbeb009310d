#!/bin/bash
# A/B timing of one build with and without an environment switch: tools/ab_env.sh VAR=value workload...
SW=$1; shift
for w in "$@"; do for rep in 1 2; do for mode in off on; do
  if [ $mode = on ]; then export $SW; else unset ${SW%%=*}; fi
  timeout -k 10 100 python bench.py --no-extras --workload $w 2>/dev/null | python -c "
import sys,json; j=json.loads(sys.stdin.read()); r=j['roofline']; print('$w', '$SW $mode', 'back-to-back', round(r['kernel_ms_avg'],4), 'isolated', round(r['kernel_ms_isolated_avg'],4))" || exit 1
done; done; done
