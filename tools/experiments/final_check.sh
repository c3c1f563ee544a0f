mkdir -p gpurun_out/r03q
timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/r03q/gpu_tests.txt 2>&1; echo "rc $?" >> gpurun_out/r03q/gpu_tests.txt; tail -3 gpurun_out/r03q/gpu_tests.txt
for m in 1 3; do
  AFE_FORCE_STEP_MODE=$m timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/r03q/gpu_tests_mode$m.txt 2>&1; echo "mode $m rc $?" >> gpurun_out/r03q/gpu_tests_mode$m.txt
  tail -2 gpurun_out/r03q/gpu_tests_mode$m.txt
done
AFE_FORCE_HOST_ARENA=1 timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/r03q/gpu_tests_host_arena.txt 2>&1; echo "rc $?" >> gpurun_out/r03q/gpu_tests_host_arena.txt; tail -2 gpurun_out/r03q/gpu_tests_host_arena.txt
bash tools/profile_r03.sh r03c > gpurun_out/profile_r03c.log 2>&1
FULL=0 bash tools/profile_r03.sh r03c_ns 131072 > gpurun_out/profile_r03c_ns.log 2>&1
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
