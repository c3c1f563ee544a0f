"""what the cache-policy hints of the one-step launches buy, by ensemble size (afe_set_cache_policy)
    python tools/cache_policy_probe.py [log2 sizes ...]
bench workload (config 4: gusts, counter noise, ticks every 2nd step), launch mode, one and two streams"""
import importlib, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
afa = importlib.import_module("agri-fly_amd")
import bench

sizes = [float(a) for a in sys.argv[1:]] or [20, 21, 21.585, 22, 22.585, 23, 24]
sync = torch.cuda.synchronize
for lg in sizes:
    n = int(round(2 ** lg / 1024)) * 1024
    e = bench.build_shard(afa, n, 0, n, 0)
    e.set_step_mode(afa.AFE_STEP_LAUNCH)
    b_mean, _ = bench.mean_bytes_per_step(e, afa, 200)
    k = max(40, min(400, int(4e7 // n) * 2))
    row = []
    for parts in (1, 2):
        e.set_split_stepping(parts)
        for pol in (0, 1, 2, 3, -1):
            e.set_cache_policy(pol)
            bench.time_steps(e, 20, 1, sync, lambda: None)
            t = bench.median([bench.time_steps(e, k, 1, sync, lambda: None) for _ in range(3)]) / k
            row.append("%s%d:%7.2f us %5.0f GB/s" % ("s" if parts == 2 else "p", pol, t * 1e6, n * b_mean / t / 1e9))
    print("n = %9d (2^%.2f, %4.0f MB/step):  " % (n, lg, n * b_mean / 1e6) + " | ".join(row), flush=True)
    e.close()
