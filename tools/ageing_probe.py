"""does a resident grid slow down gradually or at a threshold?  us per step in consecutive windows of one grid's life
(own queue, never retired)     AFE_PERSIST_AQL=1 AFE_PERSIST_REFRESH_STEPS=0 python tools/ageing_probe.py [vehicles] [window]"""
import importlib, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
afa = importlib.import_module("agri-fly_amd")
import bench
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
win = int(sys.argv[2]) if len(sys.argv) > 2 else 128
for rep in range(2):
    e = bench.build_shard(afa, n, 0, n, 0)
    e.step(1000, 64); e.sync(); e.grid_time()          # (set-up costs of the engine's first launch are not this grid's)
    ts = []
    for w in range(40):
        t0 = time.perf_counter()
        for _ in range(win): e.step(1000, 1)
        e.sync()
        ts.append((time.perf_counter() - t0) / win * 1e6)
    print("%d vehicles, windows of %d steps, us/step: " % (n, win) + " ".join("%.2f" % t for t in ts) + "   resident: %s" % e.persistent_running, flush=True)
    e.close()
