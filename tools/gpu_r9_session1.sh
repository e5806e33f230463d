#!/bin/bash
# Round-5 GPU session 1: new tests, wide halo A/B, fc_cls wgrad A/B, RoIAlign XCD A/B, bench quick, PMC of the HBM kernels, 2-rank path
set -u
TAG=${1:-s1}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
echo "== host: $(nproc) cores; $(grep -m1 'model name' /proc/cpuinfo)" | tee $OUT/host.txt
echo "== new tests"
timeout 600 python -m pytest tests/test_gpu_det_ops.py tests/test_gpu_detector.py -m gpu -q --timeout 300 -k "wide_pixel or nan_and_equal or without_gts or fork_holds or halo" > $OUT/pytest_new.log 2>&1; echo "pytest new rc=$?"
grep -E "passed|failed|error" $OUT/pytest_new.log | tail -3; grep -E "^FAILED|^ERROR|Error|assert " $OUT/pytest_new.log | head -30
echo "== halo wide A/B (bf16x6)"
timeout 300 python tools/halo_wide_ab.py bf16x6 20 > $OUT/halo_wide_ab.txt 2> $OUT/halo_wide_ab.err; echo "rc=$?"; cat $OUT/halo_wide_ab.txt; tail -3 $OUT/halo_wide_ab.err
echo "== halo wide A/B (bf16)"
timeout 300 python tools/halo_wide_ab.py bf16 20 > $OUT/halo_wide_ab_bf16.txt 2> $OUT/halo_wide_ab_bf16.err; echo "rc=$?"; cat $OUT/halo_wide_ab_bf16.txt; tail -3 $OUT/halo_wide_ab_bf16.err
echo "== fc_cls wgrad A/B"
timeout 300 python tools/fc_cls_wgrad_ab.py > $OUT/fc_cls_wgrad_ab.txt 2> $OUT/fc_cls_wgrad_ab.err; echo "rc=$?"; cat $OUT/fc_cls_wgrad_ab.txt; tail -3 $OUT/fc_cls_wgrad_ab.err
echo "== bench quick (wide auto)"
timeout 900 python bench.py --no-extras --no-cpu-baseline > $OUT/bench_quick.json 2> $OUT/bench_quick.err; echo "bench rc=$?"; tail -3 $OUT/bench_quick.err
python - <<PY
import json
d=json.loads([l for l in open('$OUT/bench_quick.json') if l.startswith('{')][-1])
print({k:d.get(k) for k in ('value','ms_per_step','ms_per_step_eager','ms_per_step_graph','launch_calibration')})
for k in sorted(d):
    if k.startswith('roofline') and isinstance(d[k], dict):
        r=d[k]; print(k, {x:r.get(x) for x in ('achieved','frac','us_per_launch','us_per_call','ms_per_launch','algorithmic_bytes','sum_of_per_roi_footprints','union_of_footprints','rois_per_level','frac_with_union_footprint')})
print(d.get('roofline_hbm_kernels_error'))
PY
echo "== bench quick (wide off), RoIAlign XCD on"
BGS_HALO_WIDE=0 BGS_ROI_XCD=1 timeout 900 python bench.py --no-extras --no-cpu-baseline > $OUT/bench_quick_off.json 2> $OUT/bench_quick_off.err; echo "bench rc=$?"; tail -3 $OUT/bench_quick_off.err
python - <<PY
import json
d=json.loads([l for l in open('$OUT/bench_quick_off.json') if l.startswith('{')][-1])
print({k:d.get(k) for k in ('value','ms_per_step','ms_per_step_eager','ms_per_step_graph','launch_calibration')})
r=d.get('roofline_roi_align') or {}; print('roi_align xcd=1', {x:r.get(x) for x in ('achieved','frac','us_per_launch')})
PY
echo "== PMC of the HBM-bound helper kernels"
bash tools/pmc_hbm_kernels.sh $TAG/pmc_hbm 2>&1 | tail -30
echo "== 2 ranks on one GPU over gloo (calibration + N=1 reference code path)"
BGS_BENCH_ONE_DEVICE=1 BGS_DIST_BACKEND=gloo BGS_BENCH_NO_DIST_GRAPH_CHILD=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 5 --warmup 2 --no-roofline > $OUT/dist2.json 2> $OUT/dist2.err; echo "dist2 rc=$?"; tail -4 $OUT/dist2.err
python - <<PY
import json
ls=[l for l in open('$OUT/dist2.json') if l.startswith('{')]
if ls:
    d=json.loads(ls[-1])
    print({k:d.get(k) for k in ('value','n_gpus','ms_per_step','launch_calibration','n1_same_invocation','weak_scaling_eff','grad_exchange_check')})
PY
du -sh $OUT
