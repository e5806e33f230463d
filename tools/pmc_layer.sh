#!/bin/bash
# PMC counters of one conv layer: bash tools/pmc_layer.sh <tag> <kernel-name-substring> <conv_layer_once args...>
# Separate --pmc passes, kernel-trace only (gpurun refuses pmc + other trace domains).
set -u
TAG=$1; shift
KEY=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
run() {  # name, counters...
  local name=$1; shift
  timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -o conv -- python $R/tools/conv_layer_once.py $ARGS > $OUT/$name.log 2> $OUT/$name.err
  echo "$name rc=$?"
}
ARGS="$*"
run sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD
run sq2 SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
run tcp TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TOTAL_READ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum
run tlb TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_UTCL1_SERIALIZATION_STALL_sum
run tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum
run ta TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_LDS_WAVEFRONTS_sum TA_FLAT_READ_WAVEFRONTS_sum TD_TD_BUSY_sum TD_TC_STALL_sum
run fetch FETCH_SIZE
run write WRITE_SIZE
python - <<PY
import csv, glob, collections
for d in ['sq','sq2','tcp','tlb','tcc','ta','fetch','write']:
    files = glob.glob('$OUT/%s/**/*counter_collection.csv' % d, recursive=True)
    if not files:
        print(d, 'no counter file'); continue
    agg = collections.defaultdict(list)
    dur = collections.defaultdict(list)
    for r in csv.DictReader(open(files[0])):
        kn = r.get('Kernel_Name','')
        if '$KEY' in kn:
            agg[(kn.split('(')[0][-44:], r.get('Counter_Name'))].append(float(r.get('Counter_Value', 0)))
    for k, v in sorted(agg.items()):
        print('%-46s %-36s n=%d avg=%.6g' % (k[0], k[1], len(v), sum(v)/len(v)))
PY
find $OUT -name "*.csv" -size +5M -delete
