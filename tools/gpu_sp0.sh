#!/bin/bash
# selectp=0 bench (+ optional profile):  tools/gpu_sp0.sh <tag> [prof]
set -u
TAG=${1:-r1u}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
timeout 900 python bench.py --selectp 0 --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_sp0.json 2> $OUT/bench_sp0.err; echo "bench rc=$?"; cat $OUT/bench_sp0.json | cut -c1-600; tail -5 $OUT/bench_sp0.err
if [ "${2:-}" = "prof" ]; then
  (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o sp0 -- python $R/bench.py --selectp 0 --steps 5 --warmup 2 --no-cpu-baseline --no-graph > $OUT/prof_bench.json 2> $OUT/prof.err; echo "rocprof rc=$?")
  python tools/prof_summary.py $(find $OUT/prof -name "*kernel_trace.csv" | head -1) > $OUT/prof_summary.md 2>/dev/null; head -45 $OUT/prof_summary.md | cut -c1-200
  find $OUT/prof -name "*kernel_trace.csv" -size +30M -delete
fi
du -sh $OUT
