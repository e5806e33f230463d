#!/bin/bash
# PMC passes (separate rocprofv3 --pmc runs, kernel-trace only) of conv_bf16s_kernel on the layer3 1x1 shape of
# X101-64x4d (M = 8400, K = Cout = 1024).  Usage: bash tools/pmc_bf16s.sh <tag>
set -u
TAG=${1:-r6_pmc_bf16s}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
pass() {
  local dir=$1 name=$2 ctr=$3; shift 3
  timeout -k 3 90 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $OUT/$dir/$name -o p -- "$@" > $OUT/$dir.$name.log 2> $OUT/$dir.$name.err
  echo "$dir/$name rc=$?"
}
group() {
  local dir=$1; shift
  mkdir -p $OUT/$dir
  pass $dir sq "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "$@"
  pass $dir sq2 "SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE" "$@"
  pass $dir sq3 "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_WAVES SQ_INSTS_SALU SQ_ACTIVE_INST_SCA" "$@"
  pass $dir ta "TA_TA_BUSY_sum TA_FLAT_READ_LDS_WAVEFRONTS_sum" "$@"
  pass $dir td "TD_TD_BUSY_sum TD_TC_STALL_sum" "$@"
  pass $dir tcp "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum" "$@"
  pass $dir tcc "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" "$@"
  pass $dir fetch "FETCH_SIZE" "$@"
}
group l3 python $R/tools/bf16s_layer_once.py 2 50 84 1024 1024 0
group l1 python $R/tools/bf16s_layer_once.py 2 200 336 256 256 0
python - <<PY
import csv, glob, collections
for d in ('l3', 'l1'):
    for f in sorted(glob.glob('$OUT/%s/*/**/*counter_collection.csv' % d, recursive=True)):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            kn = r.get('Kernel_Name', '')
            if 'conv_bf16s' in kn:
                agg[(r.get('Grid_Size'), r.get('Counter_Name'))].append(float(r.get('Counter_Value', 0)))
        for k, v in sorted(agg.items()):
            print('%-4s grid %-9s %-32s n=%d avg=%.6g' % (d, k[0], k[1], len(v), sum(v) / len(v)))
PY
find $OUT -name "*.csv" -size +5M -delete
du -sh $OUT
