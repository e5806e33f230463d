#!/bin/bash
# GPU session for the detector path: new op tests + detector tests + detector bench (+ profile).
set -u
TAG=${1:-r1g}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
if [ "${4:-}" != "notest" ]; then
echo "== pytest det ops + detector"
timeout 1500 python -m pytest tests/test_gpu_det_ops.py tests/test_gpu_detector.py -m gpu -q --timeout 900 > $OUT/pytest_det.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_det.log
tail -5 $OUT/pytest_det.log
fi
echo "== bench detector"
timeout 1200 python bench.py --workload detector --steps 10 --warmup 3 --no-extras > $OUT/bench_det.json 2> $OUT/bench_det.err; echo "bench rc=$?"; cat $OUT/bench_det.json; tail -5 $OUT/bench_det.err
if [ "${2:-}" = "prof" ]; then
  (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o det -- python $R/bench.py --workload detector --steps 5 --warmup 2 --no-cpu-baseline --no-extras --no-graph > $OUT/prof_bench.json 2> $OUT/prof.err; echo "rocprof rc=$?")
  python tools/prof_summary.py $(find $OUT/prof -name "*kernel_trace.csv" | head -1) > $OUT/prof_summary.md 2>/dev/null; head -30 $OUT/prof_summary.md | cut -c1-180
  find $OUT/prof -name "*kernel_trace.csv" -size +30M -delete
fi
if [ "${3:-}" = "infer" ]; then timeout 600 python tools/infer_time.py 20 > $OUT/infer_time.json 2> $OUT/infer.err; echo "infer rc=$?"; cat $OUT/infer_time.json; tail -3 $OUT/infer.err; fi
du -sh $OUT
