OUT=$GRAFT_REPO_ROOT/gpurun_out/r2j; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o det -- python $GRAFT_REPO_ROOT/bench.py --workload detector --steps 6 --warmup 2 --no-cpu-baseline --no-extras --no-graph --no-roofline > $OUT/prof_bench.json 2> $OUT/prof.err; echo "rocprof rc=$?"
cd $GRAFT_REPO_ROOT
python tools/prof_summary.py $(find $OUT/prof -name "*kernel_trace.csv" | head -1) 8 > $OUT/prof_summary.md; head -60 $OUT/prof_summary.md | cut -c1-200
find $OUT/prof -name "*kernel_trace.csv" -size +30M -delete
