#!/bin/bash
# Round-3 PMC passes on the FINAL tree (separate rocprofv3 --pmc runs, kernel-trace only):
#   halo  : conv3x3_halo_bfx4_kernel<2,3> on the roofline layer (FPN P2 output conv)
#   lat0  : fpn.lat0 (1x1, 256->256, M=134400, upsampled residual absent) with the 64x64 operand ring
#           (BGS_CONV1X1_BRES=0) and with the filter-resident kernel (BGS_CONV1X1_BRES=2)
#   gs    : gs_head_fused_kernel (N = 1024) HBM traffic
# Usage: bash tools/pmc_r5.sh <tag>
set -u
TAG=${1:-r5_pmc}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
pass() {  # dir, name, "counters", cmd...
  local dir=$1 name=$2 ctr=$3; shift 3
  timeout -k 3 90 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $OUT/$dir/$name -o p -- "$@" > $OUT/$dir.$name.log 2> $OUT/$dir.$name.err
  echo "$dir/$name rc=$?"
}
group() {  # dir, cmd...
  local dir=$1; shift
  mkdir -p $OUT/$dir
  pass $dir sq "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "$@"
  pass $dir sq2 "SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE" "$@"
  pass $dir ta "TA_TA_BUSY_sum TA_FLAT_READ_LDS_WAVEFRONTS_sum" "$@"
  pass $dir td "TD_TD_BUSY_sum TD_TC_STALL_sum" "$@"
  pass $dir tcc "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" "$@"
  pass $dir fetch "FETCH_SIZE" "$@"
  pass $dir write "WRITE_SIZE" "$@"
}
group halo python $R/tools/conv_p2_once.py
BGS_CONV1X1_BRES=0 group lat0_ring python $R/tools/conv_layer_once.py 2 200 336 256 256 1 1
BGS_CONV1X1_BRES=2 group lat0_bres python $R/tools/conv_layer_once.py 2 200 336 256 256 1 1
mkdir -p $OUT/gs
pass gs fetch "FETCH_SIZE" python $R/bench.py --workload gs_head --steps 20 --warmup 2 --no-graph --no-cpu-baseline
pass gs write "WRITE_SIZE" python $R/bench.py --workload gs_head --steps 20 --warmup 2 --no-graph --no-cpu-baseline
python - <<PY
import csv, glob, collections, os
keys = {'halo': ('halo_bfx4',), 'lat0_ring': ('conv_igemm_bfx_dma',), 'lat0_bres': ('conv1x1_bres',), 'gs': ('gs_head_fused', 'gs_loss_rowwave')}
for d, subs in keys.items():
    for f in sorted(glob.glob('$OUT/%s/*/**/*counter_collection.csv' % d, recursive=True)):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            kn = r.get('Kernel_Name', '')
            if any(s in kn for s in subs):
                agg[(kn.split('(')[0][-44:], r.get('Grid_Size'), r.get('Counter_Name'))].append(float(r.get('Counter_Value', 0)))
        for k, v in sorted(agg.items()):
            print('%-10s %-46s grid %-9s %-28s n=%d avg=%.5g' % (d, k[0], k[1], k[2], len(v), sum(v) / len(v)))
PY
find $OUT -name "*.csv" -size +5M -delete
du -sh $OUT
