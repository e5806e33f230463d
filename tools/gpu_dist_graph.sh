#!/bin/bash
# Launch policies of the N > 1 path on ONE GPU (1-rank RCCL group whose all-reduce really runs):
#   bash tools/gpu_dist_graph.sh <tag>
set -u
TAG=${1:-dg}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
show() { python -c "import json,sys;d=json.load(open('$1'));print(d['ms_per_step'], d.get('ms_per_step_eager'), d['config']['launch'], '|', d['config'].get('collective_backend'), '|', d['last_losses']['loss'])" 2>&1 | tail -1; }
timeout 300 python bench.py --steps 30 --warmup 5 --no-extras --no-cpu-baseline --no-roofline > $OUT/single.json 2> $OUT/single.err; echo "single rc=$? $(show $OUT/single.json)"
BGS_BENCH_SELF_GROUP=1 timeout 300 python bench.py --steps 30 --warmup 5 --no-extras --no-cpu-baseline --no-roofline > $OUT/self_eager.json 2> $OUT/self_eager.err; echo "self-group eager rc=$? $(show $OUT/self_eager.json)"; tail -2 $OUT/self_eager.err | cut -c1-200
BGS_BENCH_SELF_GROUP=1 timeout 300 python bench.py --dist-graph --steps 300 --warmup 5 --no-extras --no-cpu-baseline --no-roofline > $OUT/self_graph.json 2> $OUT/self_graph.err; echo "self-group graph rc=$? $(show $OUT/self_graph.json)"; tail -3 $OUT/self_graph.err | cut -c1-300
