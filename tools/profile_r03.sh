#!/bin/bash
# rocprofv3 passes behind profiles/r03*: run on the GPU box from the repo root (gpurun).
#   bash tools/profile_r03.sh <tag> [vehicles]
# The headline steps by ONE resident grid per synchronised block (afe_set_step_mode): the kernel trace shows one
# afe_step_persistent_kernel per block, its duration / the block's steps = time per step.  Every pass below uses
# --headline-only (blocks of exactly --steps steps, nothing else), so all but the warm-up launch serve --steps steps.
# FETCH_SIZE and WRITE_SIZE in separate passes (TCC slots); counters never share a pass with API traces.
set -u
TAG=${1:-r03}
N=${2:-1048576}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
export TMPDIR=/tmp
cd /tmp
B="$ROOT/bench.py --headline-only --vehicles $N ${BENCH_EXTRA:-}"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$TAG -- python3 $B --steps 2000 --warmup 200 > $OUT/prof_$TAG.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_k20_$TAG -- python3 $B --steps 20 --warmup 5 > $OUT/prof_k20_$TAG.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch_$TAG -- python3 $B --steps 200 --warmup 20 > $OUT/pmc_fetch_$TAG.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write_$TAG -- python3 $B --steps 200 --warmup 20 > $OUT/pmc_write_$TAG.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $OUT/pmc_sq_$TAG -- python3 $B --steps 200 --warmup 20 > $OUT/pmc_sq_$TAG.log 2>&1
if [ "${FULL:-1}" = "1" ]; then
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_full_$TAG -- python3 $ROOT/bench.py --no-cpu-baseline --steps 200 --warmup 20 --vehicles $N > $OUT/prof_full_$TAG.log 2>&1
fi
cd $ROOT
ls $OUT | grep $TAG
