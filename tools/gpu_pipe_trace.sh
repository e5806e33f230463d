#!/bin/bash
# kernel trace of the pipelined step (depth from $1, default 5) and of the sequential eager step; occupancy analysis of both
set -u
D=${1:-5}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/ptrace
mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
BGS_BENCH_PIPELINE_DEPTH=$D timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/pipe -o pipe -- python $R/bench.py --no-extras --no-cpu-baseline --no-roofline --launch pipelined --steps 20 --warmup 8 > $OUT/pipe.json 2> $OUT/pipe.err; echo "pipe rc=$?"
BGS_BENCH_NO_PIPELINE=1 timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/eager -o eager -- python $R/bench.py --no-extras --no-cpu-baseline --no-roofline --launch eager --steps 20 --warmup 5 > $OUT/eager.json 2> $OUT/eager.err; echo "eager rc=$?"
cd $R
for a in pipe eager; do
  f=$(find $OUT/$a -name "*kernel_trace.csv" | head -1)
  echo "== $a: $f"; grep -o '"ms_per_step": [0-9.]*' $OUT/$a.json | head -1
  python tools/trace_busy.py $f 8 | tee $OUT/${a}_busy.txt
done
rm -rf $OUT/eager; f=$(find $OUT/pipe -name "*kernel_trace.csv" | head -1); python - "$f" <<PY
import csv,sys
rows=sorted(csv.DictReader(open(sys.argv[1])),key=lambda r:int(r["Start_Timestamp"]))
rows=rows[-1500:]
import gzip
w=gzip.open("$OUT/pipe_tail.csv.gz","wt")
w.write("start,end,stream,name\n")
[w.write("%s,%s,%s,%s\n"%(r["Start_Timestamp"],r["End_Timestamp"],r.get("Stream_Id","?"),r["Kernel_Name"].split("(")[0][-60:])) for r in rows]
w.close()
PY
rm -rf $OUT/pipe
