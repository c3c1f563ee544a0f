#!/bin/bash
# rocprofv3 passes for the HBM-proper regime (beyond the 256 MiB Infinity Cache): bench.py --headline-only at 2^22
# vehicles per GPU.   bash tools/profile_r04_bc.sh <tag> [vehicles] [extra bench args]
# Kernel trace + stats in one pass; FETCH_SIZE and WRITE_SIZE in separate passes (never with API traces).
set -u
TAG=${1:-r04_bc}
N=${2:-4194304}
EXTRA=${3:-}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
export TMPDIR=/tmp
cd /tmp
B="$ROOT/bench.py --headline-only --vehicles $N $EXTRA ${BENCH_EXTRA:-}"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$TAG -- python3 $B --steps 200 --warmup 20 > $OUT/prof_$TAG.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch_$TAG -- python3 $B --steps 100 --warmup 10 > $OUT/pmc_fetch_$TAG.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write_$TAG -- python3 $B --steps 100 --warmup 10 > $OUT/pmc_write_$TAG.log 2>&1
cd $ROOT
ls $OUT | grep $TAG
