"""where a synchronised block of K steps spends its fixed cost: the K afe_step calls (host), afe_sync, torch.cuda.synchronize
    python tools/sync_cost_probe.py [vehicles] [K]"""
import importlib, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
afa = importlib.import_module("agri-fly_amd")
import bench
n = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
K = int(sys.argv[2]) if len(sys.argv) > 2 else 20
e = bench.build_shard(afa, n, 0, n, 0)
e.set_step_mode(afa.AFE_STEP_PERSISTENT)
e.step(1000, 200); e.sync()
pc = time.perf_counter
rows = []
for rep in range(200):
    e.sync(); torch.cuda.synchronize()
    t0 = pc()
    for _ in range(K): e.step(1000, 1)
    t1 = pc()
    e.sync()
    t2 = pc()
    torch.cuda.synchronize()
    t3 = pc()
    rows.append((t1 - t0, t2 - t1, t3 - t2, t3 - t0))
r = np.median(np.array(rows) * 1e6, axis=0)
print("%d vehicles, K = %d: step calls %.1f us, afe_sync %.1f us, torch.cuda.synchronize %.1f us, block %.1f us = %.2f us per step; resident after sync: %s"
      % (n, K, r[0], r[1], r[2], r[3], r[3] / K, e.persistent_running))
# the same without the device-wide synchronise
rows = []
for rep in range(200):
    e.sync()
    t0 = pc()
    for _ in range(K): e.step(1000, 1)
    e.sync()
    rows.append(pc() - t0)
print("   engine only (afe_sync on both sides): block %.1f us = %.2f us per step" % (np.median(rows) * 1e6, np.median(rows) * 1e6 / K))
# an idle torch.cuda.synchronize
ts = []
for rep in range(200):
    t0 = pc(); torch.cuda.synchronize(); ts.append(pc() - t0)
print("   torch.cuda.synchronize on an idle device: %.1f us; afe_sync with nothing pending: " % (np.median(ts) * 1e6), end="")
ts = []
for rep in range(200):
    t0 = pc(); e.sync(); ts.append(pc() - t0)
print("%.1f us" % (np.median(ts) * 1e6))
e.close()
