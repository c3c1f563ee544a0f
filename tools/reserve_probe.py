"""what reserving one compute unit per XCD costs the resident grid (afe_set_reserved_compute_units)
    python tools/reserve_probe.py"""
import importlib, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
afa = importlib.import_module("agri-fly_amd")
import bench
sync = torch.cuda.synchronize
for n in (1 << 20, 131072, 1 << 19):
    for r in (0, 1, 2):
        e = bench.build_shard(afa, n, 0, n, 0)
        e.set_reserved_compute_units(r)
        long = bench.median(bench.timed_blocks(e, 2000, 1, sync, lambda: None, lambda x: x, min_total_s=0.1, min_blocks=3)) / 2000 * 1e6
        k20 = bench.median(bench.timed_blocks(e, 20, 1, sync, lambda: None, lambda x: x, min_total_s=0.05, settle_s=0.0)) / 20 * 1e6
        print("%8d vehicles, %d compute unit(s) per XCD reserved: %.2f us/step in 2000-step blocks, %.2f in 20-step blocks" % (n, r, long, k20), flush=True)
        e.close()
