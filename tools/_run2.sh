mkdir -p gpurun_out/r03b
timeout 900 python -m pytest tests/test_gpu_persistent.py -x -q > gpurun_out/r03b/persist_tests.txt 2>&1; echo "rc $?" >> gpurun_out/r03b/persist_tests.txt
tail -30 gpurun_out/r03b/persist_tests.txt
timeout 600 python tools/persist_probe.py 1024 4096 16384 131072 1048576 2097152 > gpurun_out/r03b/persist_probe.txt 2>&1; echo "rc $?" >> gpurun_out/r03b/persist_probe.txt
cat gpurun_out/r03b/persist_probe.txt
# a grid cut too large on purpose: must recover by itself
AFE_PERSIST_WAVES_PER_CU=30 timeout 300 python tools/persist_probe.py 1048576 > gpurun_out/r03b/persist_stall.txt 2>&1; echo "rc $?" >> gpurun_out/r03b/persist_stall.txt
cat gpurun_out/r03b/persist_stall.txt
