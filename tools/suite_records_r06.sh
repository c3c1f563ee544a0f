#!/bin/bash
# The GPU suite as the round's records hold it: plain, under the five forced stepping / placement modes, and on the host-sanitized
# library (tools/host_sanitizers.sh).  gpurun --timeout 3000 -- 'bash tools/suite_records_r06.sh'  ->  gpurun_out/suite_r06/
cd "$(dirname "$0")/.."
OUT=gpurun_out/suite_r06; mkdir -p $OUT
{ echo "== plain"; timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -6; } > $OUT/suite.txt
cp gpurun_out/parity_ledger.json $OUT/parity_ledger.json 2>/dev/null
: > $OUT/forced_modes.txt
for m in AFE_FORCE_STEP_MODE=1 AFE_FORCE_STEP_MODE=3 AFE_FORCE_SPLIT=1 AFE_FORCE_HOST_ARENA=1 AFE_PERSIST_AQL=1; do
  { echo "== $m"; env $m timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -6; } >> $OUT/forced_modes.txt
done
bash tools/host_sanitizers.sh $OUT/host_sanitizers > /dev/null 2>&1
tail -3 $OUT/suite.txt; grep -c passed $OUT/forced_modes.txt; tail -4 $OUT/host_sanitizers/summary.txt
