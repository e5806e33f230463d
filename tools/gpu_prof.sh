#!/bin/bash
# In-step kernel trace of the headline workload only (eager launches, 6 steps): bash tools/gpu_prof.sh <tag>
set -u
TAG=${1:-prof}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
(cd /tmp && BGS_LEVEL_FORK=${BGS_LEVEL_FORK:-0} timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o det -- python $R/bench.py --workload detector --steps 6 --warmup 2 --no-cpu-baseline --no-extras --no-roofline --no-graph > $OUT/prof_bench.json 2> $OUT/prof.err; echo "rocprof rc=$?")
python tools/prof_summary.py $(find $OUT/prof -name "*kernel_trace.csv" | head -1) 8 $OUT/step_families.json "profiles/${TAG}_detector_prof_summary.md" > $OUT/prof_summary.md 2>$OUT/prof_summary.err; head -18 $OUT/prof_summary.md | cut -c1-180
for f in $(find $OUT/prof -name "*kernel_stats.csv"); do cp $f $OUT/kernel_stats.csv; done
find $OUT/prof -name "*kernel_trace.csv" -size +30M -delete
