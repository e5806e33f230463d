mkdir -p gpurun_out/r2m
timeout 900 python -m pytest tests/test_gpu_cascade.py -m gpu -q --timeout 300 > gpurun_out/r2m/pytest.log 2>&1; echo "pytest rc=$?"; tail -30 gpurun_out/r2m/pytest.log | cut -c1-400
