#!/bin/bash
# Full validation of the tree on one GPU box: GPU parity tests, smoke, default bench, rocprofv3 kernel trace.
# Usage: bash tools/gpu_full.sh <tag>
set -u
TAG=${1:-full}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
echo "== host: $(nproc) cores; $(grep -m1 'model name' /proc/cpuinfo)" | tee $OUT/host.txt
echo "== pytest -m gpu"
timeout 1200 python -m pytest tests -m gpu -q --timeout 600 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log
tail -6 $OUT/pytest_gpu.log | cut -c1-300
echo "== smoke"; timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $OUT/smoke.log; tail -3 $OUT/smoke.log | cut -c1-300
echo "== bench"
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; cut -c1-600 $OUT/bench.json; tail -3 $OUT/bench.err
echo "== rocprofv3 kernel trace (eager, 8 steps; BGS_LEVEL_FORK=0: with the small pyramid levels on a side stream the"
echo "   durations of overlapping kernels are not additive, so the per-family table is taken on one stream)"
(cd /tmp && BGS_LEVEL_FORK=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o det -- python $R/bench.py --workload detector --steps 6 --warmup 2 --no-cpu-baseline --no-extras --no-roofline --no-graph > $OUT/prof_bench.json 2> $OUT/prof.err; echo "rocprof rc=$?")
python tools/prof_summary.py $(find $OUT/prof -name "*kernel_trace.csv" | head -1) 8 $OUT/step_families.json "profiles/${TAG}_detector_prof_summary.md" > $OUT/prof_summary.md 2>$OUT/prof_summary.err; head -18 $OUT/prof_summary.md | cut -c1-180
for f in $(find $OUT/prof -name "*kernel_stats.csv"); do cp $f $OUT/kernel_stats.csv; done
find $OUT/prof -name "*kernel_trace.csv" -size +30M -delete
du -sh $OUT
