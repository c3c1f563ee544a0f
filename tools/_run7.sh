mkdir -p gpurun_out/r03h
timeout 900 python -m pytest tests/test_gpu_planner.py -x -q 2>&1 | tail -5
for spec in "whole:AFE_PLANNER_ROUNDS_FROM=100000000" "default:" "fine:AFE_PLANNER_ROUNDS_US=250,500,1000,2000,4000,8000,16000,32000" "coarse:AFE_PLANNER_ROUNDS_US=4000,16000"; do
  name=${spec%%:*}; envs=${spec#*:}
  for scene in synthetic orchard; do
    for n in 16384 65536; do
      echo -n "$name $scene: "; env $envs timeout 600 python tools/planner_probe.py $n $scene 2>&1 | grep planners
    done
  done
done > gpurun_out/r03h/planner_rounds.txt 2>&1
cat gpurun_out/r03h/planner_rounds.txt
