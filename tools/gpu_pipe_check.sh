#!/bin/bash
# Round 5: the pipelined launch policy in the default bench line (N = 1) and on the N > 1 code path (2 ranks over gloo
# on the one GPU), plus the tests that exercise side streams.
set -u
TAG=${1:-s13}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
timeout 900 python bench.py --no-extras --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; tail -2 $OUT/bench.err
python - <<PY
import json
d=json.loads([l for l in open("$OUT/bench.json") if l.startswith("{")][-1])
print({k:d.get(k) for k in ("value","ms_per_step","ms_per_step_eager","ms_per_step_graph","launch_calibration")})
print(d["config"]["launch"][:160])
print('roofline_step', d["roofline_step"]["frac"], d["roofline_step"].get("per_layer_floor",{}).get("frac"))
PY
BGS_BENCH_ONE_DEVICE=1 BGS_DIST_BACKEND=gloo BGS_BENCH_NO_DIST_GRAPH_CHILD=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 5 --warmup 2 --no-roofline > $OUT/dist2.json 2> $OUT/dist2.err; echo "dist2 rc=$?"; tail -3 $OUT/dist2.err | cut -c1-200
python - <<PY
import json
ls=[l for l in open("$OUT/dist2.json") if l.startswith("{")]
if ls:
    d=json.loads(ls[-1]); print({k:d.get(k) for k in ("value","n_gpus","ms_per_step","launch_calibration","n1_same_invocation","weak_scaling_eff")})
PY
timeout 900 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_detector.py -m gpu -q -k "fork or pipeline or training_iteration" 2>&1 | tail -4
