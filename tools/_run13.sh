mkdir -p gpurun_out/r03n
timeout 600 python tools/persist_probe.py > gpurun_out/r03n/persist_probe.txt 2>&1; tail -12 gpurun_out/r03n/persist_probe.txt
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r03n/bench_k20.json 2> gpurun_out/r03n/bench_k20.err; echo "bench rc $?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03n/bench_k20.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['steady_state'], d['north_star_shard']['us_per_step'], d['north_star_shard']['frac'])
print(d['closed_loop_on_device'])
PY
