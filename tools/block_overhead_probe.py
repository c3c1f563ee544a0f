"""what a synchronised block of K steps costs around the steps: T(K) = a + b K, resident grid vs launches
    python tools/block_overhead_probe.py [vehicles]"""
import importlib, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
afa = importlib.import_module("agri-fly_amd")
import bench
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
for name, mode in (("resident grid", afa.AFE_STEP_PERSISTENT), ("launches", afa.AFE_STEP_LAUNCH)):
    e = bench.build_shard(afa, n, 0, n, 0)
    e.set_step_mode(mode)
    e.step(1000, 100); e.sync()
    ks, ts = [1, 2, 5, 10, 20, 50, 100, 200], []
    for k in ks:
        best = []
        for rep in range(30):
            e.sync(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(k): e.step(1000, 1)
            e.sync(); torch.cuda.synchronize()
            best.append(time.perf_counter() - t0)
        ts.append(float(np.median(best)) * 1e6)
    b, a = np.polyfit(ks, ts, 1)
    print("%d vehicles, %s: T(K) = %.1f us + %.2f us x K   (" % (n, name, a, b) + ", ".join("K=%d: %.0f" % kv for kv in zip(ks, ts)) + ")", flush=True)
    e.close()
