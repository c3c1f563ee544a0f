"""afe_set_step_mode: what one resident grid buys over one launch per step, by ensemble size
(the bench's workload: gust force, IMU + noise at 500 Hz, one step per afe_step call).
    python tools/persist_probe.py [sizes...]"""
import importlib, os, sys, time
import torch  # noqa: F401
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
afa = importlib.import_module("agri-fly_amd")
import bench

sizes = [int(x) for x in sys.argv[1:]] or [4096, 65536, 131072, 262144, 524288, 1 << 20]
for n in sizes:
    steps = 4000 if n <= (1 << 18) else 2000
    row = {}
    for name, mode, parts in (("launch", afa.AFE_STEP_LAUNCH, 1), ("split", afa.AFE_STEP_LAUNCH, 2), ("persistent", afa.AFE_STEP_PERSISTENT, 1), ("resident", afa.AFE_STEP_RESIDENT, 1)):
        e = bench.build_shard(afa, n, 0, n, 0)
        e.set_split_stepping(parts)
        e.set_step_mode(mode)
        for _ in range(200): e.step(1000, 1)
        e.sync()
        best = 1e9
        for rep in range(3):
            t0 = time.perf_counter()
            for _ in range(steps): e.step(1000, 1)
            e.sync()
            best = min(best, time.perf_counter() - t0)
        t0 = time.perf_counter()
        e.step(1000, steps)          # the same steps authorised by one call (host cost out of the picture)
        e.sync()
        one_call = (time.perf_counter() - t0) / steps * 1e6
        row[name] = (best / steps * 1e6, one_call)
        b = bench.mean_bytes_per_step(e, afa, steps)[0]
        e.close()
    print("%8d vehicles: " % n + "  ".join("%s %.2f us (%.2f in one call, %.2f TB/s)" % (k, v[0], v[1], n * b / min(v) / 1e6) for k, v in row.items()), flush=True)
