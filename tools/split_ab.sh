#!/bin/bash
# dot2c split (default library) against the subtract form (libbgs_splitsub.so): planes, per-layer times, step times
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
OUT=gpurun_out/split_ab.txt
SUB=$PWD/balancedgroupsoftmax_amd/libbgs_splitsub.so
{
echo "== planes test"; timeout 600 python -m pytest tests/test_gpu_det_ops.py -q -x -m gpu -k "split_planes or bfx_error or dgrad" 2>&1 | tail -3
echo "== planes (default)"; timeout 300 python tools/split_ab.py planes
echo "== planes (subtract form)"; BGS_LIB_PATH=$SUB timeout 300 python tools/split_ab.py planes
for r in 1 2; do
echo "== layers (default) round $r"; timeout 300 python tools/split_ab.py layers | tail -1
echo "== layers (subtract form) round $r"; BGS_LIB_PATH=$SUB timeout 300 python tools/split_ab.py layers | tail -1
done
echo "== layers (default), per layer"; timeout 300 python tools/split_ab.py layers
echo "== layers (subtract form), per layer"; BGS_LIB_PATH=$SUB timeout 300 python tools/split_ab.py layers
for r in 1 2; do
echo "== step (default) round $r"; timeout 600 python tools/step_ab.py "eager:;pipe5:PIPE=5" 3 2>&1 | tail -3
echo "== step (subtract form) round $r"; BGS_LIB_PATH=$SUB timeout 600 python tools/step_ab.py "eager:;pipe5:PIPE=5" 3 2>&1 | tail -3
done
} > $OUT 2>&1
tail -60 $OUT
