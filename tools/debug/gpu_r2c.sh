mkdir -p gpurun_out/r2c
timeout 900 python tools/conv_bfx_check.py --out gpurun_out/r2c/bfx_sweep.txt > gpurun_out/r2c/bfx_check.log 2>&1
echo "bfx_check rc=$?"; grep -c " ok" gpurun_out/r2c/bfx_check.log; grep "BAD\|CORRECT\|MISMATCH\|Error\|error" gpurun_out/r2c/bfx_check.log | head
timeout 300 python tools/debug/htc_chain_metrics.py > gpurun_out/r2c/htc_metrics.txt 2>&1; cat gpurun_out/r2c/htc_metrics.txt | tail -12
timeout 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r2c/bench.json 2> gpurun_out/r2c/bench.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r2c/bench.json'))
print({k:d[k] for k in ('value','ms_per_step','ms_per_step_eager')}, d['roofline']['ms_per_launch'], {k:v['ms_per_step'] for k,v in d['also_measured'].items()})
PY
