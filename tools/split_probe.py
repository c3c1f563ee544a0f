"""afe_set_split_stepping: what stepping the two halves of an ensemble on two streams buys, by ensemble size
(the bench's workload: gust force, IMU + noise at 500 Hz, one launch per 1 ms step and per half).
    python tools/split_probe.py"""
import importlib, os, sys, time
import torch  # noqa: F401
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
afa = importlib.import_module("agri-fly_amd")
import bench

for n in (65536, 131072, 262144, 524288, 1 << 20, 2 << 20, 4 << 20):
    steps = 2000 if n <= (1 << 20) else 500
    row = []
    for parts in (1, 2):
        e = bench.build_shard(afa, n, 0, n, 0)
        e.set_step_mode(afa.AFE_STEP_LAUNCH)
        e.set_split_stepping(parts)
        for _ in range(200): e.step(1000, 1)
        e.sync()
        best = 1e9
        for rep in range(3):
            t0 = time.perf_counter()
            for _ in range(steps): e.step(1000, 1)
            e.sync()
            best = min(best, time.perf_counter() - t0)
        row.append(best / steps * 1e6)
        e.close()
    print("%8d vehicles: %.2f us per step on one stream, %.2f us split (%+.1f %%)" % (n, row[0], row[1], (row[0] / row[1] - 1) * 100), flush=True)
