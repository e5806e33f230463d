mkdir -p gpurun_out/r2t
timeout 900 python -m pytest tests -m gpu -q --timeout 300 > gpurun_out/r2t/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r2t/pytest.log | cut -c1-300
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r2t/bench.json 2> gpurun_out/r2t/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.load(open('gpurun_out/r2t/bench.json'))
print({k:d[k] for k in ('value','ms_per_step','ms_per_step_eager')}, d['roofline']['ms_per_launch'], d['roofline']['frac'])
print({k:(v.get('ms_per_step')) for k,v in d['also_measured'].items()})
PY
