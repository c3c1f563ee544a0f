"""fused launches (afe_step(dt, k), k > 1): one stream or two halves on two streams?  2^19 .. 2^21 vehicles.
   python tools/fused_split_probe.py"""
import importlib, os, sys, time
import torch  # noqa: F401
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
afa = importlib.import_module("agri-fly_amd")
import bench
for n in (1 << 19, 1 << 20, 1 << 21):
    for k in (2, 4, 10, 50):
        row = []
        for parts in (1, 2):
            e = bench.build_shard(afa, n, 0, n, 0)
            e.set_step_mode(afa.AFE_STEP_LAUNCH)
            e.set_split_stepping(parts)
            for _ in range(20): e.step(1000, k)
            e.sync()
            best = 1e9
            calls = max(20, 2000 // k)
            for rep in range(3):
                t0 = time.perf_counter()
                for _ in range(calls): e.step(1000, k)
                e.sync()
                best = min(best, time.perf_counter() - t0)
            row.append(best / (calls * k) * 1e6)
            e.close()
        print("%8d vehicles, %2d steps per call: %.2f us per step on one stream, %.2f split (%+.1f %%)" % (n, k, row[0], row[1], (row[0] / row[1] - 1) * 100), flush=True)
