"""Can the shared-world query hide under the resident grid?  Engine e steps (persistent, 10 steps authorised per cycle);
engine q -- same device, its own stream -- runs the neighbour query on a fixed gathered buffer once per cycle.
    python tools/overlap_probe.py [vehicles]"""
import importlib, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
afa = importlib.import_module("agri-fly_amd")
import bench

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
for age in (0, 3000):
    e = bench.build_shard(afa, n, 0, n, 0)
    q = bench.build_shard(afa, n, 0, n, 0)
    if age:
        e.step(1000, age); e.sync()
    xyz = torch.empty((3, n), dtype=torch.float32, device="cuda")
    e.pack_positions(xyz.data_ptr()); e.sync()
    d2 = torch.empty(n, dtype=torch.float32, device="cuda"); idx = torch.empty(n, dtype=torch.int32, device="cuda")
    q.set_neighbour_grid_refresh(1 << 30)
    q.nearest_neighbour(xyz.data_ptr(), n, d2.data_ptr(), idx.data_ptr()); q.sync()
    cycles = 60

    def run(physics, query):
        e.sync(); q.sync(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(cycles):
            if physics: e.step(1000, 10)
            if query: q.nearest_neighbour(xyz.data_ptr(), n, d2.data_ptr(), idx.data_ptr())
        e.sync(); q.sync(); torch.cuda.synchronize()
        return (time.perf_counter() - t0) / cycles * 1e6
    for _ in range(2): run(True, True)
    tp, tq, tb = min(run(True, False) for _ in range(3)), min(run(False, True) for _ in range(3)), min(run(True, True) for _ in range(3))
    print("%d vehicles, world aged %d steps: 10 steps %.0f us, one query %.0f us, both concurrently %.0f us per cycle -> %.2f of physics-only (serial would be %.2f)"
          % (n, age, tp, tq, tb, tp / tb, tp / (tp + tq)), flush=True)
    e.close(); q.close()
