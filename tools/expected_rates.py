#!/usr/bin/env python3
"""profiles/expected_rates.json from a one-GPU bench record (bench_detail.json of `python bench.py --steps 20 --warmup 5`):
what ONE MI355X delivers on the shard each rank of an N-GPU run holds, in the driver's own protocol.  `bench.py --gpus N`
prints the measured per-GPU rate beside these (scaling_check), so that a first multi-GPU run can be judged at a glance:
shards do not communicate while stepping, so the ratio should be ~1.
    python tools/expected_rates.py gpurun_out/bench_detail.json"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
d = json.load(open(sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "bench_detail.json")))
assert d["n_gpus"] == 1
n = d["config"]["vehicles_per_gpu"]
out = {"from": "bench.py --steps %d --warmup %d on one MI355X (round 6)" % (d["steps"], d["warmup"]), "steps": d["steps"],
       "noise": d["config"]["noise"][:40],
       "weak_per_gpu": {str(n): {"vsteps_per_s": d["value"], "ms_per_step": d["ms_per_step"]}},
       "strong_shard": {}}
for k, r in (d.get("strong_shard_rates_one_gpu") or {}).get("rows", {}).items():
    out["strong_shard"][k] = {"vsteps_per_s": r["vsteps_per_s"], "us_per_step": r["us_per_step_k_blocks"]}
c4 = d.get("config4_as_stated")
if c4:
    out["strong_shard"][str(c4["vehicles_per_gpu"])] = {"vsteps_per_s": c4["value"], "us_per_step": c4["ms_per_step"] * 1e3}
json.dump(out, open(os.path.join(ROOT, "profiles", "expected_rates.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
