#!/bin/bash
# Runs on the GPU box (via gpurun): the round's evidence on the tree as it is -- the GPU suite with its clause counts, rocprofv3
# kernel stats and PMC passes of six workloads (counters keyed by the hash of the kernel sources), then the default bench line
# (which reads those counters), then the parity table.   SPEECH_COMMIT=<hash> tools/round_evidence.sh r3
set -u
cd /root/repo
R=${1:-r3}
timeout -k 10 600 python -u -m pytest tests -x -q -m gpu --timeout 120 --timeout-method thread -p no:cacheprovider > gpurun_out/${R}_gputests.log 2>&1 || { tail -5 gpurun_out/${R}_gputests.log; exit 1; }
tail -3 gpurun_out/${R}_gputests.log
cp gpurun_out/parity_clauses.json gpurun_out/${R}_parity_clauses.json
rm -f gpurun_out/${R}_counters.json
bash tools/gpu_prof.sh ${R} "fbank80_100k_ragged_1-30s:fast_mel_kernel<13, 1" "mfcc13_1000x10s:fast_mel_kernel<13, 2" "mfcc_hires40_1000x10s:fast_mel_kernel<13, 2" "plp13_1000x10s:fast_mel_kernel<13, 3" "spectrogram_1000x10s:fast_spectrogram_kernel<13" "fbank80_48kHz_1000x10s:fast_mel_kernel<10, 1" > gpurun_out/${R}_prof.log 2>&1 || { tail -5 gpurun_out/${R}_prof.log; exit 1; }
tail -7 gpurun_out/${R}_prof.log
cp gpurun_out/${R}_counters.json profiles/${R}_counters.json
timeout -k 10 600 python bench.py > gpurun_out/${R}_bench.json 2> gpurun_out/${R}_bench.err || { tail -5 gpurun_out/${R}_bench.err; exit 1; }
python -c "
import json; j=json.load(open('gpurun_out/${R}_bench.json')); print(j['value'], j['ms_per_step'], j['roofline']['frac'], j['roofline'].get('counters_stale'), j.get('output_check'))"
timeout -k 10 900 python tools/parity_table.py gpurun_out/${R}_parity_table > gpurun_out/${R}_parity_table.log 2>&1 || { tail -5 gpurun_out/${R}_parity_table.log; exit 1; }
tail -3 gpurun_out/${R}_parity_table.log
