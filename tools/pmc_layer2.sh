#!/bin/bash
# Memory-path PMC counters of one conv layer, ONE counter group per pass and a short timeout per pass
# (a counter set the SDK rejects aborts rocprofv3 and leaves the child hanging until the timeout):
#   bash tools/pmc_layer2.sh <tag> <kernel-name-substring> <conv_layer_once args...>
set -u
TAG=$1; shift
KEY=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
ARGS="$*"
run() {  # name, counters...
  local name=$1; shift
  timeout -k 3 45 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -o conv -- python $R/tools/conv_layer_once.py $ARGS > $OUT/$name.log 2> $OUT/$name.err
  echo "$name rc=$?"
}
run lat TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum
run tcpa TCP_TOTAL_CACHE_ACCESSES_sum TCP_TOTAL_READ_sum
run tcps TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum
run tlb TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum
run ta TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum
run ta2 TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_LDS_WAVEFRONTS_sum
run td TD_TD_BUSY_sum TD_TC_STALL_sum
python - <<PY
import csv, glob, collections
for d in ['lat','tcpa','tcps','tlb','ta','ta2','td']:
    files = glob.glob('$OUT/%s/**/*counter_collection.csv' % d, recursive=True)
    if not files:
        print(d, 'no counter file'); continue
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(files[0])):
        kn = r.get('Kernel_Name','')
        if '$KEY' in kn:
            agg[r.get('Counter_Name')].append(float(r.get('Counter_Value', 0)))
    for k, v in sorted(agg.items()):
        print('%-40s n=%d avg=%.6g' % (k, len(v), sum(v)/len(v)))
PY
find $OUT -name "*.csv" -size +5M -delete
