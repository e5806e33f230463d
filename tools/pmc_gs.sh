#!/bin/bash
# HBM traffic of the GroupSoftmax kernels from PMC counters (separate passes, kernel-trace only).
set -u
TAG=${1:-pmc_gs}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/$C -o gs -- python $R/tools/pmc_gs_once.py > $OUT/$C.log 2> $OUT/$C.err
  echo "$C rc=$?"
done
python - <<PY
import csv, glob, collections
for C in ['FETCH_SIZE','WRITE_SIZE']:
    files = glob.glob('$OUT/%s/**/*counter_collection.csv' % C, recursive=True)
    if not files:
        print(C, 'no counter file'); continue
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(files[0])):
        name = r.get('Kernel_Name','')
        if 'gs_' in name:
            agg[(name.split('(')[0][-70:], r.get('Grid_Size'), r.get('Counter_Name'))].append(float(r.get('Counter_Value', 0)))
    for k, v in sorted(agg.items()):
        tail = v[len(v)//2:]
        print(C, k, 'n=%d' % len(v), 'avg(last half)=%.1f' % (sum(tail)/len(tail)))
PY
find $OUT -name "*.csv" -size +5M -delete
