mkdir -p gpurun_out/r03f
timeout 900 python -m pytest tests/test_gpu_sharedworld.py tests/test_gpu_counter.py tests/test_gpu_persistent.py -x -q 2>&1 | tail -3
timeout 1200 python bench.py --steps 20 --warmup 5 --no-sweep --no-cpu-baseline > gpurun_out/r03f/bench.json 2> gpurun_out/r03f/bench.err; echo "rc $?"
tail -3 gpurun_out/r03f/bench.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r03f/bench.json'))
print(json.dumps(d['shared_world'], indent=1))
PY
