mkdir -p gpurun_out/r2q
timeout 900 python -m pytest tests/test_gpu_gs.py tests/test_gpu_detector.py tests/test_gpu_e2e.py -m gpu -q --timeout 300 > gpurun_out/r2q/pytest.log 2>&1; echo "pytest rc=$?"; tail -12 gpurun_out/r2q/pytest.log | cut -c1-300
timeout 300 python bench.py --workload gs_head --no-cpu-baseline > gpurun_out/r2q/gs_head.json 2> gpurun_out/r2q/gs_head.err; echo "gs_head rc=$?"; python -c "
import json; d=json.load(open('gpurun_out/r2q/gs_head.json')); print(d['value'], d['ms_per_step'], d.get('ms_per_step_eager'), d['roofline'])"
