mkdir -p gpurun_out/r2o
timeout 900 python -m pytest tests/test_gpu_mask.py tests/test_gpu_detector.py tests/test_gpu_htc.py -m gpu -q --timeout 300 > gpurun_out/r2o/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r2o/pytest.log | cut -c1-300
