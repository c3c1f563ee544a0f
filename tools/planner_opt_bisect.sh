#!/bin/bash
# Which compiler pass makes the planner's orchard campaign fail when side_scan is inlined?  LLVM's -opt-bisect-limit=N runs
# only the first N optional pass invocations; a binary search over N between a passing and a failing build ends on the
# first invocation whose presence changes the outcome.  Runs on the GPU box (hipcc is there too): gpurun -- bash tools/planner_opt_bisect.sh
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
SRC=$ROOT/agri-fly_amd/csrc
VAR=$ROOT/agri-fly_amd/lib/variants
mkdir -p "$VAR"
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-result -I$ROOT/include ${PLANNER_DEFS:--DAFE_NI_SIDE=__forceinline__}"
OTHERS=$(ls $ROOT/agri-fly_amd/lib/obj/*.o | grep -v afe_planner.o)
try() {   # $1 = limit; returns 0 if the campaign passes
  /opt/rocm/bin/hipcc $FLAGS -mllvm -opt-bisect-limit=$1 -x hip -c $SRC/afe_planner.hip -o "$VAR/planner_bisect.o" 2> "$VAR/bisect_$1.log" || { echo "  build failed at $1"; return 2; }
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $OTHERS "$VAR/planner_bisect.o" -o "$VAR/libagrifly_engine_bisect.so" -ldl || return 2
  AGRIFLY_ENGINE_LIB="$VAR/libagrifly_engine_bisect.so" timeout 300 python -m pytest $ROOT/tests/test_gpu_planner.py -x -q -k "campaign_on_rendered" -p no:cacheprovider > "$VAR/bisect_run.log" 2>&1
}
total=$(/opt/rocm/bin/hipcc $FLAGS --cuda-device-only -mllvm -opt-bisect-limit=-1 -x hip -c $SRC/afe_planner.hip -o /dev/null 2>&1 | grep -c BISECT)
echo "device compilation: $total optional pass invocations"
lo=${LO:-0}; hi=${HI:-$total}
try $lo; echo "limit $lo: rc $?"
try $hi; echo "limit $hi: rc $?"
while [ $((hi - lo)) -gt 1 ]; do
  mid=$(((lo + hi) / 2))
  if try $mid; then lo=$mid; r=pass; else hi=$mid; r=FAIL; fi
  echo "limit $mid: $r   (bracket $lo .. $hi)"
done
echo "first invocation whose presence breaks the campaign: $hi"
grep "BISECT: running pass ($hi)" "$VAR/bisect_$hi.log" | head -3
grep "BISECT: running pass ($((hi - 1)))" "$VAR/bisect_$hi.log" | head -1
grep "BISECT: running pass ($((hi + 1)))" "$VAR/bisect_$total.log" 2>/dev/null | head -1
rm -f "$VAR"/bisect_*.log "$VAR/planner_bisect.o"
