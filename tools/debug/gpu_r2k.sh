mkdir -p gpurun_out/r2k
timeout 600 python -m pytest tests/test_gpu_e2e.py -m gpu -q -s -k "bf16 or cascade" --timeout 300 > gpurun_out/r2k/pytest.log 2>&1; echo "pytest rc=$?"; grep "bf16 cascade\|passed\|failed\|Error\|assert" gpurun_out/r2k/pytest.log | head -20
for m in bf16x6 bf16; do
timeout 300 python bench.py --cascade --selectp 3 --conv-math $m --steps 10 --warmup 3 --no-extras --no-cpu-baseline --no-roofline > gpurun_out/r2k/cascade_$m.json 2> gpurun_out/r2k/cascade_$m.err; echo "cascade $m rc=$?"; python -c "
import json; d=json.load(open('gpurun_out/r2k/cascade_$m.json')); print(d['value'], d['ms_per_step'], d['dtype'][:40], d['last_losses']['loss'])"
done
timeout 300 python bench.py --htc --selectp 3 --conv-math bf16 --steps 10 --warmup 3 --no-extras --no-cpu-baseline --no-roofline > gpurun_out/r2k/htc_bf16.json 2> gpurun_out/r2k/htc_bf16.err; echo "htc bf16 rc=$?"; python -c "
import json; d=json.load(open('gpurun_out/r2k/htc_bf16.json')); print(d['value'], d['ms_per_step'], d['last_losses']['loss'])"
