set -x
mkdir -p gpurun_out/r03a
timeout 60 tools/hostpoll_probe.bin 2048 > gpurun_out/r03a/hostpoll.txt 2>&1; echo "rc $?" >> gpurun_out/r03a/hostpoll.txt
timeout 60 tools/hostpoll_probe.bin 8192 >> gpurun_out/r03a/hostpoll.txt 2>&1; echo "rc $?" >> gpurun_out/r03a/hostpoll.txt
timeout 600 python tools/split_probe.py > gpurun_out/r03a/split_probe.txt 2>&1
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r03a/gpu_tests.txt 2>&1; echo "rc $?" >> gpurun_out/r03a/gpu_tests.txt
tail -3 gpurun_out/r03a/gpu_tests.txt; cat gpurun_out/r03a/hostpoll.txt gpurun_out/r03a/split_probe.txt
