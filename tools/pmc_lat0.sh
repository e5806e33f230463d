#!/bin/bash
# PMC counters of fpn.lat0 (1x1, 256 -> 256, 2 x 200 x 336): the M-stacked 128 x 128 kernel (default) and the 64 x 64
# operand ring (BGS_BFX_WIDE=0).  Separate --pmc passes, kernel-trace only.
set -u
TAG=${1:-pmc_lat0}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
for ARM in wide ring; do
  if [ $ARM = ring ]; then export BGS_BFX_WIDE=0; else unset BGS_BFX_WIDE; fi
  timeout -k 3 120 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES --output-format csv -d $OUT/${ARM}_sq -o conv -- python $R/tools/conv_layer_once.py 2 200 336 256 256 1 1 > $OUT/${ARM}_sq.log 2> $OUT/${ARM}_sq.err
  echo "$ARM sq rc=$?"
  timeout -k 3 120 rocprofv3 --kernel-trace --pmc TA_TA_BUSY_sum TD_TD_BUSY_sum TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE --output-format csv -d $OUT/${ARM}_ta -o conv -- python $R/tools/conv_layer_once.py 2 200 336 256 256 1 1 > $OUT/${ARM}_ta.log 2> $OUT/${ARM}_ta.err
  echo "$ARM ta rc=$?"; tail -1 $OUT/${ARM}_ta.err
done
python - <<PY
import csv, glob, collections
for d in ['wide_sq','wide_ta','ring_sq','ring_ta']:
    files = glob.glob('$OUT/%s/**/*counter_collection.csv' % d, recursive=True)
    if not files:
        print(d, 'no counter file'); continue
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(files[0])):
        kn = r.get('Kernel_Name','')
        if 'bfx' in kn and 'split_weights' not in kn:
            agg[(kn.split('(')[0][-44:], r.get('Counter_Name'))].append(float(r.get('Counter_Value', 0)))
    for k, v in sorted(agg.items()):
        print('%-8s %-46s %-28s n=%d avg=%.5g' % (d, k[0], k[1], len(v), sum(v)/len(v)))
    tr = glob.glob('$OUT/%s/**/*kernel_trace.csv' % d, recursive=True)
    if tr:
        ds = [int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in csv.DictReader(open(tr[0])) if 'bfx' in r['Kernel_Name'] and 'split_weights' not in r['Kernel_Name']]
        if ds: print('%-8s duration avg %.1f us (n=%d)' % (d, sum(ds)/len(ds)/1e3, len(ds)))
PY
find $OUT -name "*.csv" -size +5M -delete
