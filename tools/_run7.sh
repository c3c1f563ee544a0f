mkdir -p gpurun_out/r03h
for spec in "whole:AFE_PLANNER_LPT_FROM=100000000" "lpt400:" "lpt100:AFE_PLANNER_SIZING_US=100" "lpt1000:AFE_PLANNER_SIZING_US=1000" "lpt2500:AFE_PLANNER_SIZING_US=2500"; do
  name=${spec%%:*}; envs=${spec#*:}
  for scene in synthetic orchard; do
    for n in 16384 65536; do
      echo -n "$name $scene: "; env $envs timeout 600 python tools/planner_probe.py $n $scene 2>&1 | grep planners
    done
  done
done > gpurun_out/r03h/planner_lpt.txt 2>&1
cat gpurun_out/r03h/planner_lpt.txt
