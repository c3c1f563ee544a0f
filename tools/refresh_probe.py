"""does retiring a resident grid every N steps still pay?  us per step in blocks of 2000 steps (one afe_step call each, afe_sync after)
    AFE_PERSIST_AQL=0|1 AFE_PERSIST_REFRESH_STEPS=0|512 python tools/refresh_probe.py [vehicles] [block]"""
import importlib, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
afa = importlib.import_module("agri-fly_amd")
import bench
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
blk = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
e = bench.build_shard(afa, n, 0, n, 0)
e.set_step_mode(afa.AFE_STEP_PERSISTENT)
e.step(1000, 3000); e.sync()
ts = []
for rep in range(8):
    t0 = time.perf_counter()
    e.step(1000, blk); e.sync()
    ts.append((time.perf_counter() - t0) / blk * 1e6)
print("%d vehicles, blocks of %d steps, AQL=%s REFRESH=%s: us/step %s  (median %.3f)" % (n, blk, os.environ.get("AFE_PERSIST_AQL", "auto"),
      os.environ.get("AFE_PERSIST_REFRESH_STEPS", "default"), " ".join("%.2f" % t for t in ts), float(np.median(ts))), flush=True)
e.close()
