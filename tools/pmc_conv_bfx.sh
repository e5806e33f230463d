#!/bin/bash
# PMC counters of the bf16x6 conv roofline kernel (P2 layer): SQ (two passes), L2 hit/miss, FETCH/WRITE.
# Separate --pmc passes, kernel-trace only (gpurun refuses pmc + other trace domains).
set -u
TAG=${1:-pmc_bfx}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
run() {  # name, counters...
  local name=$1; shift
  timeout -k 3 120 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -o conv -- python $R/tools/conv_p2_once.py > $OUT/$name.log 2> $OUT/$name.err
  echo "$name rc=$?"
}
run sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
run sq2 SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE
run tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum
run fetch FETCH_SIZE
run write WRITE_SIZE
python - <<PY
import csv, glob, collections
for d in ['sq','sq2','tcc','fetch','write']:
    files = glob.glob('$OUT/%s/**/*counter_collection.csv' % d, recursive=True)
    if not files:
        print(d, 'no counter file'); continue
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(files[0])):
        kn = r.get('Kernel_Name','')
        if 'bfx' in kn and 'split_weights' not in kn:
            agg[(kn.split('(')[0][-40:], r.get('Counter_Name'))].append(float(r.get('Counter_Value', 0)))
    for k, v in sorted(agg.items()):
        print('%-42s %-28s n=%d avg=%.5g' % (k[0], k[1], len(v), sum(v)/len(v)))
PY
find $OUT -name "*.csv" -size +5M -delete
