mkdir -p gpurun_out/r03c
timeout 600 python -m pytest tests/test_gpu_persistent.py -x -q 2>&1 | tail -3
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r03c/bench_k20.json 2> gpurun_out/r03c/bench_k20.err; echo "rc $?"
tail -5 gpurun_out/r03c/bench_k20.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r03c/bench_k20.json'))
print({k:d[k] for k in ('value','ms_per_step','repeats','ms_per_step_min','ms_per_step_max')})
r=d['roofline']; print({k:r[k] for k in ('achieved','frac','peak_measured','frac_of_measured','kernel_us','kernel_us_min','kernel_us_max','steady_state','launch_mode')})
print(r.get('beyond_cache')); print(d['config4_as_stated']); print(d.get('north_star_shard'))
for row in d['sweep']: print(row)
print(d['closed_loop_on_device']); print(d['shared_world']); print(d['cpu_baseline'])
PY
