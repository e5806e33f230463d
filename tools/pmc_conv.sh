#!/bin/bash
# HBM traffic of the conv roofline kernel from PMC counters (separate passes, kernel-trace only).
set -u
TAG=${1:-pmc_conv}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/$C -o conv -- python $R/tools/conv_p2_once.py > $OUT/$C.log 2> $OUT/$C.err
  echo "$C rc=$?"
done
python - <<PY
import csv, glob, collections
for C in ['FETCH_SIZE','WRITE_SIZE']:
    files = glob.glob('$OUT/%s/*counter_collection.csv' % C)
    if not files:
        print(C, 'no counter file'); continue
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(files[0])):
        name = r.get('Kernel_Name','')[:48]
        if 'conv_igemm' in name or 'conv3x3_halo' in name:
            agg[(name, r.get('Grid_Size'), r.get('Counter_Name'))].append(float(r.get('Counter_Value', 0)))
    for k, v in agg.items():
        print(C, k, 'n=%d' % len(v), 'avg=%.1f' % (sum(v)/len(v)), 'min=%.1f' % min(v))
PY
find $OUT -name "*.csv" -size +5M -delete
du -sh $OUT
