#!/bin/bash
# where the spectrogram kernel's time goes: 0 = all, 1 = no feature stores, 2 = no audio loads, 3 = neither, 8 = stores land in an L2-resident window
for w in "${@:-spectrogram_1000x10s}"; do for a in 0 8 1 2 3; do SPEECH_AMD_LIB=$PWD/speech_amd/lib/libspeech_diag.so SPEECH_ABLATE=$a timeout -k 10 100 python bench.py --no-extras --no-check --steps 20 --workload $w 2>/dev/null | python -c "
import sys,json; j=json.loads(sys.stdin.read()); r=j['roofline']; print('$w ablate $a', 'sustained', round(r['kernel_ms_avg'],4), 'isolated', round(r['kernel_ms_isolated_avg'],4))" || exit 1; done; done
