#!/bin/bash
# Every rocprofv3 pass behind profiles/r06* (round 6; the r05 script with this round's tags -- every summary records kernel_sources): run on the GPU box from the repo root (gpurun), RELEASE library.
#   bash tools/profile_r06.sh [part ...]      parts: cal h20 ns bc23 perception full   (default: all)
# Counters never share a pass with API traces; FETCH_SIZE and WRITE_SIZE in separate passes (TCC slots).
#   cal         FETCH_SIZE / WRITE_SIZE against known bytes (tools/fetch_calibration.py) -> profiles/r06_fetch_calibration.*
#   h20         the headline, 2^20 vehicles: tools/profile_r03.sh r06 (the grid is launched per synchronised block on the HIP stream)
#   ns          the north-star shard, 131 072 vehicles: tools/profile_r04.sh r06_ns 131072 (own AQL queue; AFE_GRID_LOG pairs grids and steps)
#   h20c, nsc   the same two with the counter-based noise policy (bench.py --noise counter): tags r06c, r06c_ns
#   bc23        2^23 vehicles, nothing survives a step in the Infinity Cache: tools/profile_r04_bc.sh r06_bc23 8388608
#   perception  SQ counters of the planner's search kernel and the depth camera's kernel
#   full        kernel stats of the whole default bench with the driver's arguments
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
PARTS=${@:-h20 ns bc23 h20c nsc perception full}      # (cal: round 5's calibration stands, profiles/r05_fetch_calibration.json)
for part in $PARTS; do
  case $part in
    cal)
      cd /tmp
      rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_cal_fetch_r06 -- python3 $ROOT/tools/fetch_calibration.py > $OUT/pmc_cal_fetch_r06.log 2>&1
      rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_cal_write_r06 -- python3 $ROOT/tools/fetch_calibration.py > $OUT/pmc_cal_write_r06.log 2>&1
      cd $ROOT ;;
    h20)  FULL=0 bash $ROOT/tools/profile_r03.sh r06 1048576 ;;
    ns)   FULL=0 bash $ROOT/tools/profile_r04.sh r06_ns 131072 ;;
    h20c) FULL=0 BENCH_EXTRA="--noise counter" bash $ROOT/tools/profile_r03.sh r06c 1048576 ;;
    nsc)  FULL=0 BENCH_EXTRA="--noise counter" bash $ROOT/tools/profile_r04.sh r06c_ns 131072 ;;
    bc23) bash $ROOT/tools/profile_r04_bc.sh r06_bc23 8388608 ;;
    perception)
      bash $ROOT/tools/planner_pmc.sh r06 65536 orchard > $OUT/planner_pmc_r06.txt 2>&1
      bash $ROOT/tools/render_pmc.sh r06 > $OUT/render_pmc_r06.txt 2>&1 ;;
    full)
      cd /tmp
      rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_full_r06 -- python3 $ROOT/bench.py --no-cpu-baseline --steps 20 --warmup 5 > $OUT/prof_full_r06.log 2>&1
      cd $ROOT ;;
  esac
done
ls $OUT | grep r06 | head -50
# ---- the committed summaries, made on the box from the passes above (kernel_sources = the tree that ran) and brought home
# under gpurun_out/profiles_r06/ (copy them into profiles/)
cd $ROOT
NOTE="round 6: release library, rocprofv3 passes of tools/profile_r06.sh"
for part in $PARTS; do
  case $part in
    h20)  python3 tools/profile_summary_r03.py r06 "$NOTE (reference noise streams)" 1048576 148 512 reference_streams > $OUT/summary_r06.log 2>&1 ;;
    h20c) python3 tools/profile_summary_r03.py r06c "$NOTE (counter noise)" 1048576 144 512 counter no-traffic-json > $OUT/summary_r06c.log 2>&1 ;;
    ns)   python3 tools/profile_summary_r04.py r06_ns "$NOTE (reference noise streams)" 131072 148 reference_streams > $OUT/summary_r06_ns.log 2>&1 ;;
    nsc)  python3 tools/profile_summary_r04.py r06c_ns "$NOTE (counter noise)" 131072 144 counter > $OUT/summary_r06c_ns.log 2>&1 ;;
    bc23) python3 tools/profile_summary_r04_bc.py r06_bc23 "$NOTE" 8388608 148 > $OUT/summary_r06_bc23.log 2>&1 ;;
    perception) python3 tools/perception_pmc_summary.py r06 > $OUT/summary_r06_perception.log 2>&1 ;;
  esac
done
mkdir -p $OUT/profiles_r06
cp profiles/r06* profiles/traffic.json $OUT/profiles_r06/ 2>/dev/null
ls $OUT/profiles_r06
