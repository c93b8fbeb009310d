#!/bin/bash
# Runs on the GPU box (via gpurun): kernel-trace stats and PMC passes of one bench command, each pass a run of its own.
# usage: tools/profile_round.sh <tag> [workload] [extra bench args...]      -> gpurun_out/prof_<tag>/
# then:  python tools/pmc_to_json.py gpurun_out/prof_<tag> <workload> <kernel substring> profiles/r2_counters.json
set -u
TAG=${1:-r2}
WL=${2:-fbank80_100k_ragged_1-30s}
shift 2 || true
OUT=/root/repo/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python3 /root/repo/bench.py --steps 5 --warmup 1 --no-extras --workload $WL $*"
echo "$BENCH" > $OUT/command.txt
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $BENCH > $OUT/trace.log 2>&1 || { echo "trace failed"; exit 1; }
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $BENCH > $OUT/pmc_fetch.log 2>&1 || { echo "pmc fetch failed"; exit 1; }
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $BENCH > $OUT/pmc_write.log 2>&1 || { echo "pmc write failed"; exit 1; }
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq -- $BENCH > $OUT/pmc_sq.log 2>&1 || { echo "pmc sq failed"; exit 1; }
timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq2 -- $BENCH > $OUT/pmc_sq2.log 2>&1 || { echo "pmc sq2 failed"; exit 1; }
echo "profile done"
