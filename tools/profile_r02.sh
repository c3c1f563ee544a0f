#!/bin/bash
# rocprofv3 passes behind profiles/r02_*: run on the GPU box from the repo root (gpurun).
#   bash tools/profile_r02.sh <tag>          (TICK_MODE=split: SQ counters of the half-shard launches of split stepping)
# kernel trace + stats of the bench's timed cadence; FETCH_SIZE and WRITE_SIZE in separate passes
# (TCC slots); SQ counters of the step kernels at the bench cadence; kernel stats of the whole default
# bench (shared-world, perception rows, sweep).  Counters never share a pass with API traces.
set -u
TAG=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
export TMPDIR=/tmp
cd /tmp
B="$ROOT/bench.py --no-sweep --no-cpu-baseline --no-shared-world --headline-only"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$TAG -- python3 $B --steps 1000 --warmup 100 > $OUT/prof_$TAG.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch_$TAG -- python3 $B --steps 100 --warmup 10 > $OUT/pmc_fetch_$TAG.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write_$TAG -- python3 $B --steps 100 --warmup 10 > $OUT/pmc_write_$TAG.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $OUT/pmc_sq_$TAG -- python3 $ROOT/tools/tick_probe.py ${TICK_MODE:-mix} 1048576 40 > $OUT/pmc_sq_$TAG.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_full_$TAG -- python3 $ROOT/bench.py --no-cpu-baseline --steps 200 --warmup 20 > $OUT/prof_full_$TAG.log 2>&1
cd $ROOT
ls $OUT
