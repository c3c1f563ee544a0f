mkdir -p gpurun_out/r03k
for m in 1 3; do
  AFE_FORCE_STEP_MODE=$m timeout 2400 python -m pytest tests -m gpu -x -q -p no:cacheprovider > gpurun_out/r03k/gpu_tests_mode$m.txt 2>&1; echo "mode $m rc $?" >> gpurun_out/r03k/gpu_tests_mode$m.txt
  tail -6 gpurun_out/r03k/gpu_tests_mode$m.txt
done
