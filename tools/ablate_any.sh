#!/bin/bash
# tools/ablate_any.sh "<ablate values>" workload...   (SPEECH_ABLATE bits: fast_common.hpp, FastParams::ablate)
AB=$1; shift
for w in "$@"; do for a in $AB; do SPEECH_AMD_LIB=$PWD/speech_amd/lib/libspeech_diag.so SPEECH_ABLATE=$a timeout -k 10 100 python bench.py --no-extras --no-check --steps 20 --workload $w 2>/dev/null | python -c "
import sys,json; j=json.loads(sys.stdin.read()); r=j['roofline']; print('$w ablate $a', 'sustained', round(r['kernel_ms_avg'],4), 'isolated', round(r['kernel_ms_isolated_avg'],4))" || exit 1; done; done
