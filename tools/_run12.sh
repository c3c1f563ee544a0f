mkdir -p gpurun_out/r03m
AFE_FORCE_HOST_ARENA=1 timeout 3000 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/r03m/gpu_tests_host_arena.txt 2>&1; echo "rc $?" >> gpurun_out/r03m/gpu_tests_host_arena.txt
tail -12 gpurun_out/r03m/gpu_tests_host_arena.txt
