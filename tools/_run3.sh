mkdir -p gpurun_out/r03e
timeout 1200 python bench.py --steps 20 --warmup 5 > gpurun_out/r03e/bench_k20.json 2> gpurun_out/r03e/bench_k20.err; echo "rc $?"
tail -5 gpurun_out/r03e/bench_k20.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r03e/bench_k20.json'))
print({k:d[k] for k in ('value','ms_per_step','repeats','ms_per_step_min','ms_per_step_max')})
r=d['roofline']; print({k:r[k] for k in ('achieved','frac','frac_of_measured','kernel_us','steady_state','launch_mode','algorithmic_bytes_per_vehicle_step')})
print(r['per_kernel'])
print(r.get('beyond_cache')); print(d['config4_as_stated']); print(d.get('north_star_shard'))
for row in d['sweep']: print({k:(round(v,3) if isinstance(v,float) else v) for k,v in row.items() if k in ('vehicles','us_per_step','frac','stepping')}, 'launch', round(row['launches']['us_per_step'],2), 'split', round(row['split_launches']['us_per_step'],2))
print(d['closed_loop_on_device']); print(d['companions']); print(json.dumps(d['disturbance_sweep'])); print(d['shared_world']); print(d['cpu_baseline'])
PY
