mkdir -p gpurun_out/r03n
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r03n/cycle_prof -- python3 $GRAFT_REPO_ROOT/tools/cycle_probe.py > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/r03n/cycle_prof/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# find a steady-state window: the 100th query kernel onwards, print 1.2 cycles
qi = [i for i, r in enumerate(rows) if "world_query_kernel" in r["Kernel_Name"]]
a = qi[100]; b = qi[102]
t0 = int(rows[a]["Start_Timestamp"])
prev_end = None
for r in rows[a:b + 1]:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    gap = "" if prev_end is None else "gap %.1f" % ((s - prev_end) / 1e3)
    print("%8.1f %8.1f us  %-60s %s" % (s / 1e3, (e - s) / 1e3, r["Kernel_Name"].split("(")[0][-60:], gap))
    prev_end = max(prev_end or 0, e)
PY
