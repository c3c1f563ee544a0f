#!/bin/bash
# SQ counters of the RAPPIDS search kernel: how many waves are resident, what they wait on.
#   bash tools/planner_pmc.sh <tag> [planners] [orchard]
set -u
TAG=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
export TMPDIR=/tmp
cd /tmp
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY --kernel-trace --output-format csv -d $OUT/pmc_planner_$TAG -- python3 $ROOT/tools/planner_probe.py ${2:-65536} ${3:-} > $OUT/pmc_planner_$TAG.log 2>&1
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_FLAT SQ_WAVE_CYCLES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d $OUT/pmc_planner2_$TAG -- python3 $ROOT/tools/planner_probe.py ${2:-65536} ${3:-} > $OUT/pmc_planner2_$TAG.log 2>&1
cd $ROOT
python3 - <<PY
import csv, glob, collections
for d in ("pmc_planner_$TAG", "pmc_planner2_$TAG"):
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % d, recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(float))
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"][:48]][r["Counter_Name"]] += float(r["Counter_Value"])
        for k, v in acc.items():
            if "search_kernel" in k:
                print(k, dict(v))
PY
