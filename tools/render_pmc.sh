#!/bin/bash
# SQ counters of the depth-camera kernel (1024 views of the bench's orchard): what the waves spend
# their cycles on.  Run on the GPU box from the repo root:  bash tools/render_pmc.sh <tag>
set -u
TAG=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
export TMPDIR=/tmp
cd /tmp
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/pmc_render_$TAG -- python3 $ROOT/tools/render_stats_probe.py > $OUT/pmc_render_$TAG.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INST_CYCLES_SALU SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_INSTS_VMEM_WR --kernel-trace --output-format csv -d $OUT/pmc_render2_$TAG -- python3 $ROOT/tools/render_stats_probe.py > $OUT/pmc_render2_$TAG.log 2>&1
cd $ROOT
python3 - <<PY
import csv, glob, collections
for d in ("pmc_render_$TAG", "pmc_render2_$TAG"):
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % d, recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:60]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        for k, v in acc.items():
            if "render_depth_kernel" in k:
                print(k, dict(v))
PY
