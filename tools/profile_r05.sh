#!/bin/bash
# Every rocprofv3 pass behind profiles/r05*: run on the GPU box from the repo root (gpurun), RELEASE library.
#   bash tools/profile_r05.sh [part ...]      parts: cal h20 ns bc23 perception full   (default: all)
# Counters never share a pass with API traces; FETCH_SIZE and WRITE_SIZE in separate passes (TCC slots).
#   cal         FETCH_SIZE / WRITE_SIZE against known bytes (tools/fetch_calibration.py) -> profiles/r05_fetch_calibration.*
#   h20         the headline, 2^20 vehicles: tools/profile_r03.sh r05 (the grid is launched per synchronised block on the HIP stream)
#   ns          the north-star shard, 131 072 vehicles: tools/profile_r04.sh r05_ns 131072 (own AQL queue; AFE_GRID_LOG pairs grids and steps)
#   h20c, nsc   the same two with the counter-based noise policy (bench.py --noise counter): tags r05c, r05c_ns
#   bc23        2^23 vehicles, nothing survives a step in the Infinity Cache: tools/profile_r04_bc.sh r05_bc23 8388608
#   perception  SQ counters of the planner's search kernel and the depth camera's kernel
#   full        kernel stats of the whole default bench with the driver's arguments
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
PARTS=${@:-cal h20 ns bc23 h20c nsc perception full}
for part in $PARTS; do
  case $part in
    cal)
      cd /tmp
      rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_cal_fetch_r05 -- python3 $ROOT/tools/fetch_calibration.py > $OUT/pmc_cal_fetch_r05.log 2>&1
      rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_cal_write_r05 -- python3 $ROOT/tools/fetch_calibration.py > $OUT/pmc_cal_write_r05.log 2>&1
      cd $ROOT ;;
    h20)  FULL=0 bash $ROOT/tools/profile_r03.sh r05 1048576 ;;
    ns)   FULL=0 bash $ROOT/tools/profile_r04.sh r05_ns 131072 ;;
    h20c) FULL=0 BENCH_EXTRA="--noise counter" bash $ROOT/tools/profile_r03.sh r05c 1048576 ;;
    nsc)  FULL=0 BENCH_EXTRA="--noise counter" bash $ROOT/tools/profile_r04.sh r05c_ns 131072 ;;
    bc23) bash $ROOT/tools/profile_r04_bc.sh r05_bc23 8388608 ;;
    perception)
      bash $ROOT/tools/planner_pmc.sh r05 65536 orchard > $OUT/planner_pmc_r05.txt 2>&1
      bash $ROOT/tools/render_pmc.sh r05 > $OUT/render_pmc_r05.txt 2>&1 ;;
    full)
      cd /tmp
      rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_full_r05 -- python3 $ROOT/bench.py --no-cpu-baseline --steps 20 --warmup 5 > $OUT/prof_full_r05.log 2>&1
      cd $ROOT ;;
  esac
done
ls $OUT | grep r05 | head -50
