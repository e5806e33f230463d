#!/bin/bash
# two ranks sharing ONE GPU over gloo (the artificial N > 1 code-path check): which change made the non-pipelined arms slow?
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; OUT=$R/gpurun_out/${1:-s15}; mkdir -p $OUT
for E in "A=0" "BGS_STEM_FUSED=0" "BGS_ROI_XCD=0" "BGS_STEM_FUSED=0 BGS_ROI_XCD=0 BGS_GS_MERGE_PF=0"; do
  env $E BGS_BENCH_NO_PIPELINE=1 BGS_BENCH_NO_N1_REFERENCE=1 BGS_BENCH_ONE_DEVICE=1 BGS_DIST_BACKEND=gloo BGS_BENCH_NO_DIST_GRAPH_CHILD=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 3 --warmup 1 --no-roofline > $OUT/d.json 2> $OUT/d.err
  python - <<PY
import json
ls=[l for l in open("$OUT/d.json") if l.startswith("{")]
d=json.loads(ls[-1]) if ls else {}
c=d.get("launch_calibration") or {}
print("$E", "ms_per_step", d.get("ms_per_step"), {k:v for k,v in c.items() if k.endswith("_ms")})
PY
done
