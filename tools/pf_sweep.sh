#!/bin/bash
# the L2 prefetch distance of the mel kernels (FastParams::pf_dist), swept in one process-per-value on one box with libspeech_diag.so:
# tools/pf_sweep.sh "<distances>" workload...      (0 = prefetch off)
DS=$1; shift
for w in "$@"; do for rep in 1 2; do for d in $DS; do SPEECH_AMD_LIB=$PWD/speech_amd/lib/libspeech_diag.so SPEECH_PF_DIST=$d timeout -k 10 100 python bench.py --no-extras --no-check --steps 20 --workload $w 2>/dev/null | python -c "
import sys,json; j=json.loads(sys.stdin.read()); r=j['roofline']; print('$w pf_dist $d', 'sustained', round(r['kernel_ms_avg'],4), 'isolated', round(r['kernel_ms_isolated_avg'],4))" || exit 1; done; done; done
