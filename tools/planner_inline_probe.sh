#!/bin/bash
# Which helper's inlining breaks the planner?  Builds variants of the library in which ONE of the wave-cooperative helpers
# of afe_planner.hip is force-inlined (the others stay out of line), on the CPU box:   bash tools/planner_inline_probe.sh build
# and runs the orchard campaign against each on the GPU box:                            bash tools/planner_inline_probe.sh run
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
SRC=$ROOT/agri-fly_amd/csrc
VAR=$ROOT/agri-fly_amd/lib/variants
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-result -I$ROOT/include"
if [ "${1:-build}" = "build" ]; then
  mkdir -p $VAR
  OTHERS=$(ls $ROOT/agri-fly_amd/lib/obj/*.o | grep -v afe_planner.o)
  for v in MASK RING SIDE CORNER INFLATE ALL SIDE_W2 SIDE_W3; do
    if [ $v = SIDE_W2 ]; then D="-DAFE_NI_SIDE=__forceinline__ -DAFE_PLANNER_WAVES=2"
    elif [ $v = SIDE_W3 ]; then D="-DAFE_NI_SIDE=__forceinline__ -DAFE_PLANNER_WAVES=3"
    elif [ $v = ALL ]; then D="-DAFE_NI_MASK=__forceinline__ -DAFE_NI_RING=__forceinline__ -DAFE_NI_SIDE=__forceinline__ -DAFE_NI_CORNER=__forceinline__ -DAFE_NI_INFLATE=__forceinline__"
    else D="-DAFE_NI_$v=__forceinline__"; fi
    ( /opt/rocm/bin/hipcc $FLAGS $D -x hip -c $SRC/afe_planner.hip -o $VAR/planner_$v.o && \
      /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $OTHERS $VAR/planner_$v.o -o $VAR/libagrifly_engine_$v.so -ldl && echo "built $v" ) &
  done
  wait
else
  for v in ${VARIANTS:-MASK RING SIDE CORNER INFLATE ALL SIDE_W2 SIDE_W3}; do
    echo "== $v inlined"
    AGRIFLY_ENGINE_LIB=$VAR/libagrifly_engine_$v.so timeout 600 python -m pytest $ROOT/tests/test_gpu_planner.py -x -q -k "campaign_on_rendered or matches_oracle or blocked" -p no:cacheprovider 2>&1 | tail -4
  done
fi
