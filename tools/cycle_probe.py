"""Where the host's time goes in one cycle of the shared-world loop (ten steps, gather, neighbour query) at 2^20 vehicles:
host microseconds per call with the device far behind (nothing waits), and the device's own time per cycle.
   python tools/cycle_probe.py"""
import importlib, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
afa = importlib.import_module("agri-fly_amd")
import bench
n = 1 << 20
e = bench.build_shard(afa, n, 0, n, 0)
e.set_step_mode(afa.AFE_STEP_AUTO)      # ten steps per call: the engine fuses them
comm = afa.Comm(afa.Comm.unique_id(), 0, 1, device=0)
e.set_neighbour_grid_refresh(16); e.set_neighbour_sort_reuse(8)
xyz = torch.empty((3, n), dtype=torch.float32, device="cuda")
d2 = torch.empty(n, dtype=torch.float32, device="cuda"); idx = torch.empty(n, dtype=torch.int32, device="cuda")
def cycle(t):
    a = time.perf_counter(); e.step(1000, 10)
    b = time.perf_counter(); e.gather_positions(comm, xyz.data_ptr())
    c = time.perf_counter(); e.nearest_neighbour(xyz.data_ptr(), n, d2.data_ptr(), idx.data_ptr())
    d = time.perf_counter()
    t[0] += b - a; t[1] += c - b; t[2] += d - c
for _ in range(20): cycle([0, 0, 0])
e.sync()
t = [0.0, 0.0, 0.0]; k = 200
t0 = time.perf_counter()
for _ in range(k): cycle(t)
t_host = time.perf_counter() - t0
e.sync()
t_all = time.perf_counter() - t0
print("per cycle: host %.1f us in afe_step(10), %.1f us in afe_gather_positions, %.1f us in afe_nearest_neighbour; host loop %.1f us, with the device drained %.1f us"
      % (t[0] / k * 1e6, t[1] / k * 1e6, t[2] / k * 1e6, t_host / k * 1e6, t_all / k * 1e6))
# the bench's own bracketing: 40 cycles between two synchronisations, three times
for label, reuse in (("sorted every query", 1), ("order kept, sorted every 8th", 8)):
    e.set_neighbour_sort_reuse(reuse)
    for _ in range(10): cycle([0, 0, 0])
    ts = []
    for rep in range(5):
        e.sync(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(40): cycle([0, 0, 0])
        e.sync(); torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / 40 * 1e6)
    print("%s: %s us per cycle in blocks of 40" % (label, ", ".join("%.1f" % x for x in ts)))
ts = []
for rep in range(5):
    e.sync(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(40): e.step(1000, 10)
    e.sync(); torch.cuda.synchronize()
    ts.append((time.perf_counter() - t0) / 40 * 1e6)
print("ten steps alone: %s us per cycle" % ", ".join("%.1f" % x for x in ts))
