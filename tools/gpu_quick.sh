#!/bin/bash
# Quick GPU run of a pytest selection:  tools/gpu_quick.sh <tag> <pytest args...>
set -u
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
timeout 1200 python -m pytest "$@" -q --timeout 600 > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest.log
tail -40 $OUT/pytest.log
