mkdir -p gpurun_out/r03n
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r03n/world_prof -- python3 $GRAFT_REPO_ROOT/tools/world_probe.py 20 > $GRAFT_REPO_ROOT/gpurun_out/r03n/world_probe.txt 2>&1
cd $GRAFT_REPO_ROOT
cat gpurun_out/r03n/world_probe.txt | grep -v "^W\|rocprof" | head -20
f=$(find gpurun_out/r03n/world_prof -name "*kernel_stats.csv" | head -1)
grep "world_" $f | cut -c1-60,200-400 | head -12
python3 - <<PY
import csv,sys
f="$f"
for r in csv.DictReader(open(f)):
    if "world_" in r["Name"]:
        print(r["Name"].split("(")[0][-40:], r["Calls"], r["AverageNs"], r["MinNs"], r["MaxNs"])
PY
