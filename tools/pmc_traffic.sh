#!/bin/bash
# HBM traffic of the group-softmax kernel from PMC counters (separate passes, kernel-trace only).
set -u
TAG=${1:-pmc}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/$C -o gs -- python $R/bench.py --workload gs_head --steps 20 --warmup 2 --no-graph --no-cpu-baseline > $OUT/$C.json 2> $OUT/$C.err
  echo "$C rc=$?"
  ls $OUT/$C
done
python - <<PY
import csv, glob, collections
for C in ['FETCH_SIZE','WRITE_SIZE']:
    files = glob.glob('$OUT/%s/*counter_collection.csv' % C)
    if not files:
        print(C, 'no counter file'); continue
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(files[0])):
        name = r.get('Kernel_Name','')[:60]
        if 'gs_loss_rowwave' in name or 'gs_prepare' in name or 'copyBuffer' in name:
            agg[(name, r.get('Grid_Size'), r.get('Counter_Name'))].append(float(r.get('Counter_Value', 0)))
    for k, v in agg.items():
        print(C, k, 'n=%d' % len(v), 'avg=%.1f' % (sum(v)/len(v)))
PY
du -sh $OUT
