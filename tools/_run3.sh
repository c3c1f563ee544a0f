mkdir -p gpurun_out/r03i
timeout 1500 python bench.py --steps 20 --warmup 5 > gpurun_out/r03i/bench_k20.json 2> gpurun_out/r03i/bench_k20.err; echo "rc $?"
tail -3 gpurun_out/r03i/bench_k20.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r03i/bench_k20.json'))
print({k:d[k] for k in ('value','ms_per_step','repeats','ms_per_step_min','ms_per_step_max')})
r=d['roofline']; print({k:r[k] for k in ('achieved','frac','frac_of_measured','kernel_us','steady_state','launch_mode','algorithmic_bytes_per_vehicle_step','traffic')})
print(r['per_kernel']); print(r['peak_measured'])
print(r.get('beyond_cache')); print(d['config4_as_stated']); print(d.get('north_star_shard'))
for row in d['sweep']: print({k:(round(v,3) if isinstance(v,float) else v) for k,v in row.items() if k in ('vehicles','us_per_step','frac','stepping','vsteps_per_s')}, 'launch', round(row['launches']['us_per_step'],2), 'split', round(row['split_launches']['us_per_step'],2), 'f2', round(row['fused2']['us_per_step'],2),'f50', round(row['fused50']['us_per_step'],2))
print(d['closed_loop_on_device']); print(json.dumps(d['companions'])); print([ (b['sigma_N'][1], round(b['rms_xy_m'],2), round(b['gust_only_closed_form_m'],2)) for b in d['disturbance_sweep']['bins']], d['disturbance_sweep']['wall_s'])
p=d['perception_rows']; print(p['closed_perception_loop_frame']); print({k:v for k,v in p['depth_camera'].items() if k!='roofline'}); print({k:v for k,v in p['rappids_planner'].items() if k!='roofline'})
print(json.dumps(d['shared_world'])); print(d['cpu_baseline'])
PY
