#!/bin/bash
# One GPU-box session: smoke + GPU parity tests + bench + rocprofv3 kernel trace.
# Usage (from the repo root on the GPU box):  bash tools/gpu_session.sh [tag]
set -u
TAG=${1:-r1}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
echo "== host: $(nproc) cores; $(grep -m1 'model name' /proc/cpuinfo)" | tee $OUT/host.txt
rocm-smi --showproductname 2>/dev/null | head -8 >> $OUT/host.txt
echo "== smoke" ; timeout 600 python __graft_entry__.py > $OUT/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $OUT/smoke.log
tail -3 $OUT/smoke.log
echo "== pytest -m gpu (DPP build)"
timeout 1200 python -m pytest tests -m gpu -q -x --timeout 600 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log
tail -15 $OUT/pytest_gpu.log
echo "== pytest -m gpu (ds_bpermute build, subset)"
BGS_LIB_VARIANT=nodpp timeout 600 python -m pytest tests/test_gpu_gs.py -m gpu -q -x --timeout 600 -k "fixtures or selftest" > $OUT/pytest_gpu_nodpp.log 2>&1; echo "pytest nodpp rc=$?" | tee -a $OUT/pytest_gpu_nodpp.log
tail -3 $OUT/pytest_gpu_nodpp.log
echo "== bench"
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; cat $OUT/bench.json; tail -3 $OUT/bench.err
BGS_LIB_VARIANT=nodpp timeout 600 python bench.py --no-cpu-baseline > $OUT/bench_nodpp.json 2> $OUT/bench_nodpp.err; cat $OUT/bench_nodpp.json
echo "== rocprofv3 kernel trace"
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o bench -- python $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline > $OUT/prof_bench.json 2> $OUT/prof.err; echo "rocprof rc=$?")
find $OUT/prof -type f | head; for f in $(find $OUT/prof -name "*kernel_stats.csv"); do cut -c1-160 $f | head -25; done
# keep the merge-back small
find $OUT/prof -name "*kernel_trace.csv" -size +20M -delete
du -sh $OUT
