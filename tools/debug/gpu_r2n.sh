mkdir -p gpurun_out/r2n
timeout 900 python -m pytest tests -m gpu -q --timeout 300 > gpurun_out/r2n/pytest.log 2>&1; echo "pytest rc=$?"; tail -30 gpurun_out/r2n/pytest.log | cut -c1-300
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r2n/smoke.log 2>&1; echo "smoke rc=$?"; tail -3 gpurun_out/r2n/smoke.log
