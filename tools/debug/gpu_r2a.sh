set -x
mkdir -p gpurun_out/r2a
timeout 900 python tools/conv_bfx_check.py --out gpurun_out/r2a/bfx_sweep.txt > gpurun_out/r2a/bfx_check.log 2>&1
echo "bfx_check rc=$?"
tail -5 gpurun_out/r2a/bfx_check.log
timeout 600 python -m pytest tests -m gpu -q -x --timeout 300 > gpurun_out/r2a/pytest_bfx.log 2>&1
echo "pytest rc=$?"
tail -15 gpurun_out/r2a/pytest_bfx.log
timeout 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r2a/bench_bfx.json 2> gpurun_out/r2a/bench_bfx.err
echo "bench rc=$?"
head -c 600 gpurun_out/r2a/bench_bfx.json
