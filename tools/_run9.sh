mkdir -p gpurun_out/r03j
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 20 --warmup 5 --no-sweep --no-cpu-baseline > gpurun_out/r03j/bench_torchrun.json 2> gpurun_out/r03j/bench_torchrun.err; echo "rc $?"
tail -3 gpurun_out/r03j/bench_torchrun.err
python - <<'PY'
import json
lines=[l for l in open('gpurun_out/r03j/bench_torchrun.json') if l.strip()]
print(len(lines), "line(s)")
d=json.loads(lines[-1])
print({k:d[k] for k in ('value','ms_per_step','repeats','n_gpus','scaling')}, d['config4_as_stated']['value'], d['shared_world']['rccl_ranks'], d['shared_world']['worlds']['lattice_as_started']['fraction_query_in_stream'])
PY
