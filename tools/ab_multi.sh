#!/bin/bash
# A/B/C... timing of several builds of libspeech.so on the same box, alternating: tools/ab_multi.sh libA.so libB.so ... -- workload...
LIBS=(); while [ "$1" != "--" ]; do LIBS+=("$1"); shift; done; shift
for w in "$@"; do for rep in 1 2; do for lib in "${LIBS[@]}"; do
  SPEECH_AMD_LIB=$PWD/$lib timeout -k 10 100 python bench.py --no-extras --workload $w | python -c "
import sys,json; j=json.loads(sys.stdin.read()); r=j['roofline']; print('$w', '$lib', 'back-to-back', round(r['kernel_ms_avg'],4), 'isolated', round(r['kernel_ms_isolated_avg'],4))" || exit 1
done; done; done
