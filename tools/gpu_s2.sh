#!/bin/bash
set -u
TAG=${1:-s2}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
echo "== full GPU suite"
timeout 1500 python -m pytest tests -m gpu -q --timeout 600 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"
grep -E "passed|failed|error" $OUT/pytest_gpu.log | tail -3; grep -E "^FAILED|^ERROR" $OUT/pytest_gpu.log | head -30
echo "== step A/B: wide halo"
timeout 600 python tools/step_ab.py "v4:HALO_WIDE=0;wide:HALO_WIDE=1" 6 20 > $OUT/step_ab_wide.txt 2> $OUT/step_ab_wide.err; echo "rc=$?"; cat $OUT/step_ab_wide.txt; tail -2 $OUT/step_ab_wide.err
echo "== step A/B: wide halo without forks"
timeout 600 python tools/step_ab.py "v4nofork:HALO_WIDE=0,BGS_LEVEL_FORK=0;widenofork:HALO_WIDE=1,BGS_LEVEL_FORK=0" 4 20 > $OUT/step_ab_wide_nofork.txt 2> $OUT/step_ab_wide_nofork.err; echo "rc=$?"; cat $OUT/step_ab_wide_nofork.txt; tail -2 $OUT/step_ab_wide_nofork.err
echo "== step A/B: RoIAlign XCD"
timeout 600 python tools/step_ab.py "roi0:BGS_ROI_XCD=0;roi1:BGS_ROI_XCD=1" 6 20 > $OUT/step_ab_roi.txt 2> $OUT/step_ab_roi.err; echo "rc=$?"; cat $OUT/step_ab_roi.txt; tail -2 $OUT/step_ab_roi.err
echo "== step A/B selectp=0: wide halo"
timeout 600 python tools/step_ab.py "v4:HALO_WIDE=0;wide:HALO_WIDE=1" 4 10 0 > $OUT/step_ab_wide_sp0.txt 2> $OUT/step_ab_wide_sp0.err; echo "rc=$?"; cat $OUT/step_ab_wide_sp0.txt; tail -2 $OUT/step_ab_wide_sp0.err
du -sh $OUT
