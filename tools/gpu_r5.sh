#!/bin/bash
# Round-3 GPU session:  tools/gpu_r5.sh <tag> [tests] [bench] [dist] [prof] [pmc]   (any subset, in this order)
set -u
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
echo "== host: $(nproc) cores; $(grep -m1 'model name' /proc/cpuinfo)" | tee $OUT/host.txt
for what in "$@"; do
case $what in
tests)
  timeout 1500 python -m pytest tests -m gpu -q --timeout 600 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log
  grep -E "passed|failed|error" $OUT/pytest_gpu.log | tail -3; grep -E "^FAILED|^ERROR" $OUT/pytest_gpu.log | head -20
  grep -E "reproduced by the HIP RPN|max \|diff\||cascade X101" $OUT/pytest_gpu.log | head -40
  ;;
newtests)
  timeout 1200 python -m pytest tests/test_gpu_gs.py tests/test_gpu_e2e.py -m gpu -q -s --timeout 600 -k "head_step or shipped or x101 or fused" > $OUT/pytest_new.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_new.log
  grep -E "passed|failed|error" $OUT/pytest_new.log | tail -3; grep -E "^FAILED|^ERROR|Error|assert" $OUT/pytest_new.log | head -30
  grep -E "reproduced by the HIP RPN|max \|diff\||cascade X101" $OUT/pytest_new.log | head -40
  ;;
smoke)
  timeout 600 python __graft_entry__.py > $OUT/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $OUT/smoke.log; tail -3 $OUT/smoke.log
  ;;
bench)
  timeout 1200 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; tail -3 $OUT/bench.err
  python - <<PY
import json
d=json.loads([l for l in open('$OUT/bench.json') if l.startswith('{')][-1])
print({k:d.get(k) for k in ('value','ms_per_step','ms_per_step_eager')})
for k in ('roofline','roofline_step','roofline_gs_loss','roofline_gs_loss_rowwave','roofline_gs_loss_n65536'):
    r=d.get(k) or {}
    print(k, {x:r.get(x) for x in ('achieved','frac','ms_per_launch','us_per_launch','kernel')})
print('gs_head', d.get('gs_head'))
cb=d.get('cpu_baseline') or {}
print('cpu_baseline', {x:cb.get(x) for x in ('value','kind','cores','threads_tried')})
print({k:(v.get('ms_per_step'), v.get('error')) for k,v in (d.get('also_measured') or {}).items()})
PY
  ;;
benchquick)
  timeout 900 python bench.py --no-extras > $OUT/bench_quick.json 2> $OUT/bench_quick.err; echo "bench rc=$?"; tail -3 $OUT/bench_quick.err
  python - <<PY
import json
d=json.loads([l for l in open('$OUT/bench_quick.json') if l.startswith('{')][-1])
print({k:d.get(k) for k in ('value','ms_per_step','ms_per_step_eager')})
for k in ('roofline','roofline_step','roofline_gs_loss','roofline_gs_loss_rowwave','roofline_gs_loss_n65536'):
    r=d.get(k) or {}
    print(k, {x:r.get(x) for x in ('achieved','frac','ms_per_launch','us_per_launch','kernel')})
print('gs_head', d.get('gs_head'))
cb=d.get('cpu_baseline') or {}
print('cpu_baseline', {x:cb.get(x) for x in ('value','kind','cores','threads_tried')})
PY
  ;;
dist)
  # the N > 1 code path on the one GPU: two ranks over gloo (diagnostics, dist-graph children + fallback)
  BGS_BENCH_ONE_DEVICE=1 BGS_DIST_BACKEND=gloo timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 5 --warmup 2 --no-roofline > $OUT/dist2.json 2> $OUT/dist2.err; echo "dist2 rc=$?"; tail -4 $OUT/dist2.err
  python - <<PY
import json
ls=[l for l in open('$OUT/dist2.json') if l.startswith('{')]
if ls:
    d=json.loads(ls[-1])
    print({k:d.get(k) for k in ('value','n_gpus','ms_per_step','rccl_ranks','launch_policy','ms_per_step_by_rank','grad_exchange_check','allreduce_us','dist_graph_policy')})
PY
  # the RCCL-in-graph policy with a 1-rank group whose all-reduce really runs
  BGS_BENCH_SELF_GROUP=1 timeout 600 python bench.py --child --dist-graph --steps 10 --warmup 3 --no-extras --no-cpu-baseline --no-roofline > $OUT/selfgroup_graph.json 2> $OUT/selfgroup_graph.err; echo "selfgroup rc=$?"; tail -2 $OUT/selfgroup_graph.err; cut -c1-400 $OUT/selfgroup_graph.json
  ;;
prof)
  (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o det -- python $R/bench.py --workload detector --steps 6 --warmup 2 --no-cpu-baseline --no-extras --no-graph --no-roofline > $OUT/prof_bench.json 2> $OUT/prof.err; echo "rocprof rc=$?")
  python tools/prof_summary.py $(find $OUT/prof -name "*kernel_trace.csv" | head -1) 6 $OUT/step_families.json "profiles/${TAG}_detector_prof_summary.md" > $OUT/prof_summary.md 2>/dev/null; head -24 $OUT/prof_summary.md | cut -c1-170
  find $OUT/prof -name "*kernel_trace.csv" -size +30M -delete
  ;;
profsp0)
  (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof0 -o det -- python $R/bench.py --workload detector --selectp 0 --steps 5 --warmup 2 --no-cpu-baseline --no-extras --no-graph --no-roofline > $OUT/prof0_bench.json 2> $OUT/prof0.err; echo "rocprof rc=$?")
  python tools/prof_summary.py $(find $OUT/prof0 -name "*kernel_trace.csv" | head -1) 5 > $OUT/prof0_summary.md 2>/dev/null; head -40 $OUT/prof0_summary.md | cut -c1-170
  find $OUT/prof0 -name "*kernel_trace.csv" -size +30M -delete
  ;;
gsprof)
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/gsprof -o gs -- python $R/bench.py --workload gs_head --steps 50 --warmup 5 --no-graph --no-cpu-baseline > $OUT/gsprof_bench.json 2> $OUT/gsprof.err; echo "rocprof rc=$?")
  for f in $(find $OUT/gsprof -name "*kernel_stats.csv"); do cut -c1-150 $f | head -14; done
  ;;
gstests)
  timeout 600 python -m pytest tests/test_gpu_gs.py -m gpu -q --timeout 300 > $OUT/pytest_gs.log 2>&1; echo "pytest gs rc=$?" | tee -a $OUT/pytest_gs.log
  grep -E "passed|failed|error" $OUT/pytest_gs.log | tail -3; grep -E "^FAILED|^ERROR|Error|assert " $OUT/pytest_gs.log | head -20
  ;;
gsab)
  timeout 600 python tools/gs_head_ab.py 3 ${GSAB_VARIANTS:-0,1,2,3,4,5} > $OUT/gs_head_ab.txt 2> $OUT/gs_head_ab.err; echo "gsab rc=$?"; cat $OUT/gs_head_ab.txt; tail -3 $OUT/gs_head_ab.err
  ;;
gspmc)
  # HBM traffic of the head kernel: FETCH_SIZE / WRITE_SIZE in separate rocprofv3 --pmc passes (kernel-trace only)
  for C in FETCH_SIZE WRITE_SIZE; do
    (cd /tmp && timeout -k 3 150 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/gspmc/$C -o p -- python $R/bench.py --workload gs_head --steps 20 --warmup 2 --no-graph --no-cpu-baseline > $OUT/gspmc_$C.log 2> $OUT/gspmc_$C.err; echo "pmc $C rc=$?")
  done
  python - <<PY
import csv, glob, collections
for f in sorted(glob.glob('$OUT/gspmc/**/*counter_collection.csv', recursive=True)):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        kn = r.get('Kernel_Name', '')
        if 'gs_head_multi' in kn or 'gs_head_fused' in kn or 'gs_loss_rowwave' in kn:
            agg[(kn.split('(')[0][-44:], r.get('Grid_Size'), r.get('Counter_Name'))].append(float(r.get('Counter_Value', 0)))
    for k, v in sorted(agg.items()):
        print('%-46s grid %-9s %-12s n=%d avg=%.6g' % (k[0], k[1], k[2], len(v), sum(v) / len(v)))
PY
  find $OUT/gspmc -name "*.csv" -size +5M -delete
  ;;
*) echo "running: $what"; bash -c "$what" ;;
esac
done
du -sh $OUT
