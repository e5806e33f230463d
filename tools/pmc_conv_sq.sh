#!/bin/bash
# SQ counters of the conv roofline kernel (one pass, kernel-trace only).
set -u
TAG=${1:-pmc_sq}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/sq -o conv -- python $R/tools/conv_p2_once.py > $OUT/sq.log 2> $OUT/sq.err
echo "rc=$?"; tail -2 $OUT/sq.err
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d $OUT/sq2 -o conv -- python $R/tools/conv_p2_once.py > $OUT/sq2.log 2> $OUT/sq2.err
echo "rc=$?"; tail -2 $OUT/sq2.err
python - <<PY
import csv, glob, collections
for d in ['sq','sq2']:
    files = glob.glob('$OUT/%s/*counter_collection.csv' % d)
    if not files:
        print(d, 'no counter file'); continue
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(files[0])):
        if 'conv_igemm' in r.get('Kernel_Name',''):
            agg[r.get('Counter_Name')].append(float(r.get('Counter_Value', 0)))
    for k, v in sorted(agg.items()):
        print('%-28s n=%d avg=%.4g' % (k, len(v), sum(v)/len(v)))
PY
find $OUT -name "*.csv" -size +5M -delete
