mkdir -p gpurun_out/r03d
timeout 900 python -m pytest tests/test_gpu_counter.py tests/test_gpu_persistent.py -x -q > gpurun_out/r03d/counter_tests.txt 2>&1; echo "rc $?" >> gpurun_out/r03d/counter_tests.txt
tail -40 gpurun_out/r03d/counter_tests.txt
