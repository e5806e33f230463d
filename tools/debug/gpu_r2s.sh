mkdir -p gpurun_out/r2s
timeout 900 python tools/conv_bfx_check.py --out gpurun_out/r2s/bfx_sweep.txt > gpurun_out/r2s/bfx_check.log 2>&1
echo "bfx_check rc=$?"; grep -c " ok" gpurun_out/r2s/bfx_check.log; grep "BAD\|CORRECT\|MISMATCH\|Error\|error" gpurun_out/r2s/bfx_check.log | head
timeout 600 python -m pytest tests/test_gpu_det_ops.py tests/test_gpu_gs.py -m gpu -q --timeout 300 > gpurun_out/r2s/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r2s/pytest.log | cut -c1-300
timeout 300 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline > gpurun_out/r2s/bench.json 2> gpurun_out/r2s/bench.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r2s/bench.json'))
print({k:d[k] for k in ('value','ms_per_step','ms_per_step_eager')})
PY
