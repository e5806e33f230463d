mkdir -p gpurun_out/r2i
bash tools/pmc_conv_bfx.sh r2i_pmc > gpurun_out/r2i/pmc.txt 2>&1; tail -25 gpurun_out/r2i/pmc.txt
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r2i/bench.json 2> gpurun_out/r2i/bench.err; echo "bench rc=$?"; tail -3 gpurun_out/r2i/bench.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r2i/bench.json'))
print({k:d[k] for k in ('value','ms_per_step','ms_per_step_eager','dtype')})
print(d['roofline']); print(d.get('roofline_f32_mfma_kernel'))
print({k:(v.get('ms_per_step'), v.get('conv_math')) for k,v in d['also_measured'].items()})
print(d['cpu_baseline']['threads_tried'], d.get('cpu_baseline_1thread'), str(d.get('cpu_baseline_detector'))[:300])
PY
BGS_BENCH_ONE_DEVICE=1 BGS_DIST_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 5 --warmup 2 --no-extras --no-cpu-baseline --no-roofline > gpurun_out/r2i/bench_2rank_gloo_onegpu.json 2> gpurun_out/r2i/bench2.err; echo "2-rank rc=$?"; tail -3 gpurun_out/r2i/bench2.err; head -c 400 gpurun_out/r2i/bench_2rank_gloo_onegpu.json
