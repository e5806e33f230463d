#!/bin/bash
# SQ / GRBM counters of the fused stem kernel (two passes, kernel-trace only).
set -u
TAG=${1:-pmc_stem}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
timeout -k 3 200 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA --output-format csv -d $OUT/p1 -o k -- python $R/tools/stem_once.py > $OUT/p1.log 2> $OUT/p1.err; echo "p1 rc=$?"
timeout -k 3 200 rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/p2 -o k -- python $R/tools/stem_once.py > $OUT/p2.log 2> $OUT/p2.err; echo "p2 rc=$?"
python - <<PY
import csv, glob, collections
for d in ['p1', 'p2']:
    cc = glob.glob('$OUT/%s/**/*counter_collection.csv' % d, recursive=True)
    kt = glob.glob('$OUT/%s/**/*kernel_trace.csv' % d, recursive=True)
    if not cc:
        print(d, 'missing'); continue
    dur = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) for r in csv.DictReader(open(kt[0])) if 'stem_conv7x7' in r['Kernel_Name']][2:]
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(cc[0])):
        if 'stem_conv7x7' in r.get('Kernel_Name', ''):
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
    print(d, 'avg %.1f us' % (sum(dur) / len(dur) / 1e3), {k: '%.4g' % (sum(v[2:]) / len(v[2:])) for k, v in sorted(agg.items())})
PY
find $OUT -name "*.csv" -size +5M -delete
