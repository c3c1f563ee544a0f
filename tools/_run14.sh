mkdir -p gpurun_out/r03n
timeout 900 python -m pytest tests/test_gpu_persistent.py tests/test_gpu_counter.py tests/test_gpu_host_visible.py tests/test_gpu_sharedworld.py -x -q -p no:cacheprovider 2>&1 | tail -3
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r03n/bench_k20b.json 2> gpurun_out/r03n/bench_k20b.err; echo "bench rc $?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03n/bench_k20b.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'])
for k,w in d['shared_world']['worlds'].items():
    print(k, {x:w[x] for x in w if 'vsteps' in x or 'fraction' in x or x.endswith('_ms')})
print(d['disturbance_sweep']['wall_s'], d['disturbance_sweep']['vsteps_per_s'])
print(d['config1_host_in_loop'])
PY
