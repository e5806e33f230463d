mkdir -p gpurun_out/r2l
timeout 900 python -m pytest tests/test_gpu_compat.py tests/test_gpu_gs.py tests/test_gpu_mask.py tests/test_gpu_htc.py tests/test_gpu_detector.py -m gpu -q --timeout 300 > gpurun_out/r2l/pytest.log 2>&1; echo "pytest rc=$?"; tail -25 gpurun_out/r2l/pytest.log | cut -c1-300
