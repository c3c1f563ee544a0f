mkdir -p gpurun_out/r03g
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r03g/gpu_tests.txt 2>&1; echo "rc $?" >> gpurun_out/r03g/gpu_tests.txt
tail -25 gpurun_out/r03g/gpu_tests.txt
