#!/bin/bash
# LDS bank conflicts of a mel-kind kernel by phase: counters of the kernel cut off after the FFT (SPEECH_ABLATE=4), after the
# power tile (32), after the mel contraction (64) and whole (0); the differences belong to the phases in between.
# usage (on the GPU box): tools/lds_by_phase.sh workload      -> gpurun_out/lds_phase_<workload>.txt
set -u
WL=${1:-mfcc13_1000x10s}
OUT=/root/repo/gpurun_out/lds_phase_$WL
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for a in ${ABLATES:-4 32 64 0}; do
  export SPEECH_ABLATE=$a SPEECH_AMD_LIB=/root/repo/speech_amd/lib/libspeech_diag.so
  timeout -s KILL 120 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VALU --output-format csv -d $OUT/pmc_a$a -- python3 /root/repo/bench.py --steps 5 --warmup 1 --no-extras --no-check --workload $WL > $OUT/a$a.log 2>&1 || { echo "pass $a failed"; exit 1; }
done
python3 /root/repo/tools/pmc_summary.py $OUT fast_mel_kernel > /root/repo/gpurun_out/lds_phase_$WL.txt
rm -rf $OUT
cat /root/repo/gpurun_out/lds_phase_$WL.txt
