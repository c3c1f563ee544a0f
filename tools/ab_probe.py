"""A/B of two builds of the library on the bench workload: us per step of the headline stepping at the sizes given, both
noise policies, long (400-step) and short (20-step) blocks.
    AGRIFLY_ENGINE_LIB=<variant.so> python tools/ab_probe.py [vehicles ...]"""
import importlib, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import bench
afa = importlib.import_module("agri-fly_amd")
sizes = [int(x) for x in sys.argv[1:]] or [4096, 131072, 262144, 1 << 20]
sync = torch.cuda.synchronize
for exact in (True, False):
    bench.HEADLINE_EXACT_STREAMS = exact
    for n in sizes:
        e, row = bench.shard_row(afa, n, 0, sync, lambda: None, lambda x: x, 400, min_total_s=0.1)
        e.close()
        e, rk = bench.shard_row(afa, n, 0, sync, lambda: None, lambda x: x, 20, min_total_s=0.1)
        e.close()
        print("%s  %-18s %8d vehicles: %7.3f us/step in 400-step blocks, %7.3f in 20-step blocks" %
              (os.environ.get("AGRIFLY_ENGINE_LIB", "default")[-28:], "reference streams" if exact else "counter noise", n, row["us_per_step"], rk["us_per_step"]), flush=True)
