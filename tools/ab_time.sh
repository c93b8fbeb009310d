#!/bin/bash
# A/B timing of two builds of libspeech.so on the same box: tools/ab_time.sh libA.so libB.so workload...
A=$1; B=$2; shift 2
for w in "$@"; do for rep in 1 2; do for lib in $A $B; do
  SPEECH_AMD_LIB=$PWD/$lib timeout -k 10 100 python bench.py --no-extras --workload $w | python -c "
import sys,json; j=json.loads(sys.stdin.read()); r=j['roofline']; print('$w', '$lib', 'back-to-back', round(r['kernel_ms_avg'],4), 'isolated', round(r['kernel_ms_isolated_avg'],4))" || exit 1
done; done; done
