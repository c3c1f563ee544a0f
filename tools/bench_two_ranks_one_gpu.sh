#!/bin/bash
# The N > 1 control flow of bench.py on a ONE-GPU box: two ranks share device 0 and meet over gloo (AFE_BENCH_ONE_DEVICE;
# RCCL refuses two ranks on one device, hence --no-shared-world).  The two resident grids take turns on the device, so
# the numbers mean nothing -- what is checked is that both ranks get through, rank 0 prints ONE line and its bookkeeping
# (n_gpus, vehicles per rank, the strong-scaling row) is right.
#   bash tools/bench_two_ranks_one_gpu.sh
set -u
cd "$(dirname "$0")/.."
export AFE_BENCH_ONE_DEVICE=1 HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29541 \
  bench.py --gpus 2 --steps 20 --warmup 5 --vehicles 131072 --no-shared-world > /tmp/two_ranks.out 2> /tmp/two_ranks.err
rc=$?
echo "exit code $rc; stdout lines: $(wc -l < /tmp/two_ranks.out)"
python3 - <<'PY'
import json
lines = [l for l in open("/tmp/two_ranks.out").read().splitlines() if l.strip()]
assert len(lines) == 1, lines
d = json.loads(lines[0])
assert d["n_gpus"] == 2 and d["scaling"] == "weak", d
s = d["config4_as_stated"]
assert s["n_gpus"] == 2 and s["vehicles_per_gpu"] == 524288 and s["vehicles_total"] == 1048576 and s["scaling"] == "strong", s
print("one line; n_gpus", d["n_gpus"], "value %.3g" % d["value"], "| strong row:", s["vehicles_per_gpu"], "per rank, value %.3g" % s["value"])
PY
tail -3 /tmp/two_ranks.err
