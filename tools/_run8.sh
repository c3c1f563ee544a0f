timeout 900 python -m pytest tests/test_gpu_persistent.py tests/test_gpu_counter.py -x -q 2>&1 | tail -5
timeout 600 python tools/persist_probe.py 4096 131072 1048576 2>&1 | grep vehicles
