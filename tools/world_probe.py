"""Times the shared-world neighbour query on the bench's world (2^20 vehicles) for kernel / host-path A/B work:
   AGRIFLY_ENGINE_LIB=<variant.so> python tools/world_probe.py"""
import importlib, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
afa = importlib.import_module("agri-fly_amd")
import bench
n = 1 << 20
e = bench.build_shard(afa, n, 0, n, 0)
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 20): e.step(1000, 1)
xyz = torch.empty((3, n), dtype=torch.float32, device="cuda")
d2 = torch.empty(n, dtype=torch.float32, device="cuda"); idx = torch.empty(n, dtype=torch.int32, device="cuda")
e.pack_positions(xyz.data_ptr())
for refresh in (1, 16):
    e.set_neighbour_grid_refresh(refresh)
    for _ in range(3): e.nearest_neighbour(xyz.data_ptr(), n, d2.data_ptr(), idx.data_ptr())
    e.sync()
    a, b = e.event(), e.event(); e.record(a); t0 = time.perf_counter()
    for _ in range(32): e.nearest_neighbour(xyz.data_ptr(), n, d2.data_ptr(), idx.data_ptr())
    t_submit = time.perf_counter() - t0
    e.record(b); ms = e.elapsed_ms(a, b) / 32
    print("refresh every %2d queries: %.3f ms per query on the stream, %.3f ms of host time to submit one" % (refresh, ms, t_submit / 32 * 1e3))
print(e.neighbour_grid_info())
# the cell order kept over several queries (afe_set_neighbour_sort_reuse) while the ensemble flies on
e.set_neighbour_grid_refresh(16)
for reuse in (1, 4, 8, 16):
    e.set_neighbour_sort_reuse(reuse)
    for _ in range(3): e.nearest_neighbour(xyz.data_ptr(), n, d2.data_ptr(), idx.data_ptr())
    e.sync()
    a, b = e.event(), e.event(); tq = 0.0
    for _ in range(32):
        e.step(1000, 10); e.pack_positions(xyz.data_ptr())
        e.record(a); e.nearest_neighbour(xyz.data_ptr(), n, d2.data_ptr(), idx.data_ptr()); e.record(b); e.sync()
        tq += e.elapsed_ms(a, b)
    print("sort every %2d queries, ten steps of flight between queries: %.3f ms per query on the stream  %s" % (reuse, tq / 32, e.neighbour_grid_info()))
e.set_neighbour_sort_reuse(1)
e.set_neighbour_grid_refresh(16)
for cs in (6.0, 8.0, 10.0, 11.0, 12.0, 14.0, 17.0):
    for _ in range(3): e.nearest_neighbour(xyz.data_ptr(), n, d2.data_ptr(), idx.data_ptr(), cell_size=cs)
    e.sync()
    a, b = e.event(), e.event(); e.record(a)
    for _ in range(32): e.nearest_neighbour(xyz.data_ptr(), n, d2.data_ptr(), idx.data_ptr(), cell_size=cs)
    e.record(b); ms = e.elapsed_ms(a, b) / 32
    print("cell %.1f m: %.3f ms per query  %s" % (cs, ms, e.neighbour_grid_info()))
# host cost of submitting one query to an idle stream (no back-pressure from the queue)
e.set_neighbour_grid_refresh(1 << 20)
hs = []
for _ in range(20):
    e.sync()
    t0 = time.perf_counter()
    e.nearest_neighbour(xyz.data_ptr(), n, d2.data_ptr(), idx.data_ptr())
    hs.append(time.perf_counter() - t0)
e.sync()
print("host time to submit one query to an idle stream: median %.1f us" % (np.median(hs) * 1e6))
