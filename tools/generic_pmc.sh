#!/bin/bash
# Counters of the generic kernels (run on the GPU box): one rocprofv3 --pmc pass over tools/generic_time.py, per-kernel means printed.
cd /tmp && export TMPDIR=/tmp
OUT=/root/repo/gpurun_out/generic_pmc
rm -rf $OUT; mkdir -p $OUT
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq -- python3 /root/repo/tools/generic_time.py > $OUT/log.txt 2>&1 || { echo "pmc failed"; tail -5 $OUT/log.txt; exit 1; }
python3 - <<'PY'
import csv, glob, collections
per = collections.defaultdict(lambda: collections.defaultdict(lambda: collections.defaultdict(float)))
for path in glob.glob('/root/repo/gpurun_out/generic_pmc/pmc_sq/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(path)):
        if 'generic_frames_kernel' in r['Kernel_Name']:
            per[r['Kernel_Name'][:90]][r['Dispatch_Id']][r['Counter_Name']] += float(r['Counter_Value'])
for k, d in per.items():
    acc = collections.defaultdict(list)
    for dd in d.values():
        for c, v in dd.items(): acc[c].append(v)
    m = {c: sum(v) / len(v) for c, v in acc.items()}
    cyc = m['GRBM_GUI_ACTIVE'] / 8
    print(k, 'launches', len(d))
    print('   cycles %.3g valu_busy %.3f lds_busy %.3f wait_inst %.3f wait_lds %.3f valu insts/launch %.4g lds insts %.4g' % (cyc, m['SQ_ACTIVE_INST_VALU'] * 4 / 1024 / cyc, m['SQ_ACTIVE_INST_LDS'] * 4 / 1024 / cyc if 'SQ_ACTIVE_INST_LDS' in m else -1, m['SQ_WAIT_INST_ANY'] / m['SQ_WAVE_CYCLES'], m['SQ_WAIT_INST_LDS'] / m['SQ_WAVE_CYCLES'], m['SQ_INSTS_VALU'], m['SQ_INSTS_LDS']))
PY
rm -rf $OUT/pmc_sq
