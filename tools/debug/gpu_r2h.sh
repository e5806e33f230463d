mkdir -p gpurun_out/r2h
timeout 900 python -m pytest tests -m gpu -q --timeout 300 > gpurun_out/r2h/pytest.log 2>&1
echo "pytest rc=$?"
tail -40 gpurun_out/r2h/pytest.log | cut -c1-400
timeout 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r2h/bench.json 2> gpurun_out/r2h/bench.err
echo "bench rc=$?"
python - <<'PY'
import json
d=json.load(open('gpurun_out/r2h/bench.json'))
print({k:d[k] for k in ('value','ms_per_step','ms_per_step_eager')}, d['roofline'], {k:v['ms_per_step'] for k,v in d['also_measured'].items()})
PY
