mkdir -p gpurun_out/r03m
timeout 900 python -m pytest tests/test_gpu_host_visible.py -x -q -p no:cacheprovider > gpurun_out/r03m/hv_tests.txt 2>&1; echo "rc $?" >> gpurun_out/r03m/hv_tests.txt
tail -15 gpurun_out/r03m/hv_tests.txt
timeout 600 tools/host_loop_probe.bin 3000 > gpurun_out/r03m/host_loop_probe.txt 2>&1; echo "rc $?" >> gpurun_out/r03m/host_loop_probe.txt
cat gpurun_out/r03m/host_loop_probe.txt
timeout 900 python -m pytest tests/test_gpu_persistent.py tests/test_gpu_counter.py -x -q -p no:cacheprovider 2>&1 | tail -3
