#!/bin/bash
# rocprofv3 passes behind profiles/r04*: run on the GPU box from the repo root (gpurun).
#   bash tools/profile_r04.sh <tag> [vehicles]
# The resident grid lives on the engine's own AQL queue and survives afe_sync: a kernel-trace row is one grid from its
# dispatch to the moment it parked (an idle host, or an entry point that needs the stream), serving however many steps
# were authorised meanwhile.  AFE_GRID_LOG makes the engine write (vehicles, workers, steps served, device ns) per grid
# in dispatch order; tools/profile_summary_r04.py lays the two side by side.
# FETCH_SIZE and WRITE_SIZE in separate passes (TCC slots); counters never share a pass with API traces.
set -u
TAG=${1:-r04}
N=${2:-1048576}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
export TMPDIR=/tmp
cd /tmp
B="$ROOT/bench.py --headline-only --vehicles $N ${BENCH_EXTRA:-}"
rm -f $OUT/gridlog_*_$TAG.csv
AFE_GRID_LOG=$OUT/gridlog_k20_$TAG.csv rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_k20_$TAG -- python3 $B --steps 20 --warmup 5 > $OUT/prof_k20_$TAG.log 2>&1
AFE_GRID_LOG=$OUT/gridlog_$TAG.csv rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$TAG -- python3 $B --steps 2000 --warmup 200 > $OUT/prof_$TAG.log 2>&1
AFE_GRID_LOG=$OUT/gridlog_fetch_$TAG.csv rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch_$TAG -- python3 $B --steps 200 --warmup 20 > $OUT/pmc_fetch_$TAG.log 2>&1
AFE_GRID_LOG=$OUT/gridlog_write_$TAG.csv rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write_$TAG -- python3 $B --steps 200 --warmup 20 > $OUT/pmc_write_$TAG.log 2>&1
AFE_GRID_LOG=$OUT/gridlog_sq_$TAG.csv rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $OUT/pmc_sq_$TAG -- python3 $B --steps 200 --warmup 20 > $OUT/pmc_sq_$TAG.log 2>&1
if [ "${FULL:-1}" = "1" ]; then
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_full_$TAG -- python3 $ROOT/bench.py --no-cpu-baseline --steps 20 --warmup 5 --vehicles $N > $OUT/prof_full_$TAG.log 2>&1
fi
cd $ROOT
ls $OUT | grep $TAG
