#!/bin/bash
# HBM traffic (FETCH_SIZE / WRITE_SIZE, separate rocprofv3 --pmc passes with --kernel-trace only, as
# MI355X_MICROARCH.md prescribes) of the HBM-bound helper kernels of SURVEY 8(d): RoIAlign forward / backward,
# _merge_score (R = 1000, 65,536), IoU + assignment.  Usage: bash tools/pmc_hbm_kernels.sh <tag>
set -u
TAG=${1:-pmc_hbm}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  timeout -k 3 400 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/$C -o k -- python $R/tools/kernel_once.py 10 ${KERNELS:-} > $OUT/$C.log 2> $OUT/$C.err
  echo "$C rc=$?"; tail -1 $OUT/$C.log
done
python - <<PY
import csv, glob, collections, json, re
pat = re.compile(r'(roi_align_nhwc_kernel<[^>]*>|roi_align_fwd_grid_kernel<[^>]*>|gs_merge_rowwave_kernel|gs_merge_wavepriv_kernel<[^>]*>|iou_gtmax_kernel|iou_assign_kernel)')
res = collections.defaultdict(dict)
for C in ['FETCH_SIZE', 'WRITE_SIZE']:
    for f in glob.glob('$OUT/%s/**/*counter_collection.csv' % C, recursive=True):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            m = pat.search(r.get('Kernel_Name', ''))
            if m and r.get('Counter_Name') == C:
                agg[(m.group(1).replace(' ', ''), r.get('Grid_Size'))].append(float(r.get('Counter_Value', 0)))
        for (name, grid), v in sorted(agg.items()):
            v = v[len(v) // 3:]                       # drop the first third (cold caches)
            res['%s grid=%s' % (name, grid)][C] = dict(n=len(v), avg_kb=sum(v) / len(v))
            print('%-48s grid %-9s %-11s n=%d avg=%.1f KB' % (name, grid, C, len(v), sum(v) / len(v)))
json.dump(res, open('$OUT/pmc_hbm_kernels.json', 'w'), indent=1)
PY
find $OUT -name "*.csv" -size +8M -delete
du -sh $OUT
