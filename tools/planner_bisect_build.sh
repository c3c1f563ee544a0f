#!/bin/bash
# Round 4: variants of the planner with side_scan force-inlined (the build whose orchard campaign fails at the kernel's
# 128-register budget), each with ONE thing changed, to find what the failure depends on.
#   bash tools/planner_bisect_build.sh            (CPU box: builds agri-fly_amd/lib/variants/libagrifly_engine_<V>.so)
#   bash tools/planner_bisect_build.sh run        (GPU box: the orchard campaign against each)
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
SRC=$ROOT/agri-fly_amd/csrc
VAR=$ROOT/agri-fly_amd/lib/variants
FLAGS="-std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-result -I$ROOT/include"
S="-DAFE_NI_SIDE=__forceinline__"
if [ "${1:-build}" = "build" ]; then
  mkdir -p "$VAR"
  OTHERS=$(ls $ROOT/agri-fly_amd/lib/obj/*.o | grep -v afe_planner.o)
  build() { v=$1; shift; ( /opt/rocm/bin/hipcc $FLAGS "$@" -x hip -c $SRC/afe_planner.hip -o "$VAR/planner_$v.o" 2> "$VAR/build_$v.log" && \
      /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $OTHERS "$VAR/planner_$v.o" -o "$VAR/libagrifly_engine_$v.so" -ldl && echo "built $v" || { echo "FAILED $v"; tail -3 "$VAR/build_$v.log"; } ) & }
  build O1 -O1 $S
  build O2 -O2 $S
  build NOAGPR -O3 $S -mllvm -amdgpu-spill-vgpr-to-agpr=0
  build NOUNROLL -O3 $S -fno-unroll-loops
  wait
  build VOLATILE -O3 $S -DAFE_SHRINK_QUAL=volatile
  build NODPP -O3 $S -mllvm -amdgpu-dpp-combine=false
  build NOSCALARLD -O3 $S -mllvm -amdgpu-scalarize-global-loads=false
  build NOINLFN -O3 $S -fno-inline-functions
  wait
  for o in "$VAR"/planner_*.o; do rm -f "$o"; done
else
  for so in "$VAR"/libagrifly_engine_*.so; do
    v=$(basename "$so" .so); v=${v#libagrifly_engine_}
    echo "== $v"
    AGRIFLY_ENGINE_LIB=$so timeout 600 python -m pytest $ROOT/tests/test_gpu_planner.py -x -q -k "campaign_on_rendered or matches_oracle or blocked" -p no:cacheprovider 2>&1 | tail -2
  done
fi
