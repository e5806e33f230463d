#!/bin/bash
# Round-5 closing validation: full GPU suite, smoke, default bench, kernel trace, PMC of the HBM-bound kernels.
set -u
TAG=${1:-r9z}
R=${GRAFT_REPO_ROOT:-$(pwd)}
bash tools/gpu_full.sh $TAG
bash tools/pmc_hbm_kernels.sh $TAG/pmc_hbm 2>&1 | tail -14
bash tools/pmc_stem.sh $TAG/pmc_stem 2>&1 | tail -3
