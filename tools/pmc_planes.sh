#!/bin/bash
# PMC counters of the planes-in-LDS 1x1 kernel (conv1x1_planes_bfx_kernel) beside the default dispatch it replaces
# (BGS_BFX_PLANES=0: the M-stacked 128 x 128 kernel) on one layer: bash tools/pmc_planes.sh <tag> [N H W Cin Cout]
# (default fpn.lat0: 2 200 336 256 256).  Separate --pmc passes, kernel-trace only.
set -u
TAG=${1:-pmc_planes}
shift || true
SHAPE=${*:-2 200 336 256 256}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
for ARM in planes wide; do
  if [ $ARM = wide ]; then export BGS_BFX_PLANES=0; else export BGS_BFX_PLANES=2; fi
  timeout -k 3 120 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES --output-format csv -d $OUT/${ARM}_sq -o conv -- python $R/tools/conv_layer_once.py $SHAPE 1 1 > $OUT/${ARM}_sq.log 2> $OUT/${ARM}_sq.err
  echo "$ARM sq rc=$?"
  timeout -k 3 120 rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_LDS --output-format csv -d $OUT/${ARM}_sq2 -o conv -- python $R/tools/conv_layer_once.py $SHAPE 1 1 > $OUT/${ARM}_sq2.log 2> $OUT/${ARM}_sq2.err
  echo "$ARM sq2 rc=$?"
  timeout -k 3 120 rocprofv3 --kernel-trace --pmc TA_TA_BUSY_sum TD_TD_BUSY_sum TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE --output-format csv -d $OUT/${ARM}_ta -o conv -- python $R/tools/conv_layer_once.py $SHAPE 1 1 > $OUT/${ARM}_ta.log 2> $OUT/${ARM}_ta.err
  echo "$ARM ta rc=$?"; tail -1 $OUT/${ARM}_ta.err
done
python - <<PY
import csv, glob, collections
for d in ['planes_sq','planes_sq2','planes_ta','wide_sq','wide_sq2','wide_ta']:
    files = glob.glob('$OUT/%s/**/*counter_collection.csv' % d, recursive=True)
    if not files:
        print(d, 'no counter file'); continue
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(files[0])):
        kn = r.get('Kernel_Name','')
        if 'bfx' in kn and 'split_weights' not in kn:
            agg[(kn.split('(')[0][-44:], r.get('Counter_Name'))].append(float(r.get('Counter_Value', 0)))
    for k, v in sorted(agg.items()):
        print('%-10s %-46s %-28s n=%d avg=%.5g' % (d, k[0], k[1], len(v), sum(v)/len(v)))
    tr = glob.glob('$OUT/%s/**/*kernel_trace.csv' % d, recursive=True)
    if tr:
        ds = [int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in csv.DictReader(open(tr[0])) if 'bfx' in r['Kernel_Name'] and 'split_weights' not in r['Kernel_Name']]
        if ds: print('%-10s duration avg %.1f us (n=%d)' % (d, sum(ds)/len(ds)/1e3, len(ds)))
PY
find $OUT -name "*.csv" -size +5M -delete
